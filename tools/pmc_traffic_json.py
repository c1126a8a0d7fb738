#!/usr/bin/env python
"""profiles/pmc_traffic.json from the per-kernel counter table written by tools/pmc_summary.py (bench.py reads it to fill
roofline.traffic).  usage: python tools/pmc_traffic_json.py gpurun_out/pmc_bench.txt profiles/pmc_traffic.json <batch> <precision>

The file is stamped with the fingerprint of the kernel sources (bench.kernel_sources_sha), the git commit and the bench
arguments it was measured with; bench.py reports `traffic` only when all of them match the run."""
import json
import os
import re
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    txt = open(sys.argv[1]).read()
    out = {}
    for b in re.split(r"\n(?=\S)", txt):
        lines = b.strip().split("\n")
        m = re.match(r"void conv_dma_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+), 0(?:, (\w+))?(?:, (\w+))?(?:, (\w+))?(?:, (\w+))?(?:, (\w+))?>", lines[0])
        if not m:
            continue
        x3, spread, pipe, mx, swp = (m.group(i) in ("true", "1") for i in (7, 8, 9, 10, 11))
        c = {}
        for l in lines[1:]:
            q = re.match(r"\s+(\S+)\s+mean/dispatch\s+([\d.]+)\s+dispatches (\d+)", l)
            if q:
                c[q.group(1)] = (float(q.group(2)), int(q.group(3)))
        f, w = c["FETCH_SIZE"][0], c["WRITE_SIZE"][0]
        gui = c["GRBM_GUI_ACTIVE"][0] / 8.0
        key = ("conv_dma_kernel<%s,mx%s> f16x2" % (",".join(m.groups()[:6]), ",swp" if swp else (",pipe" if pipe else "")) if mx else
               "conv_dma_kernel<%s%s>%s" % (",".join(m.groups()[:6]), (",spread" if spread else "") + (",pipe" if pipe else ""), " f16x3" if x3 else ""))
        out[key] = {
            "hbm_bytes_per_launch": round((2 * f + w) * 1024),
            "FETCH_SIZE_KB_raw": f, "WRITE_SIZE_KB": w, "dispatches": c["FETCH_SIZE"][1],
            "l2_hit_rate": round(c["TCC_HIT_sum"][0] / (c["TCC_HIT_sum"][0] + c["TCC_MISS_sum"][0]), 4),
            "mfma_busy_frac_of_cycles": round(c["SQ_VALU_MFMA_BUSY_CYCLES"][0] / (gui * 1024), 4),
            "lds_bank_conflict_cycles": c["SQ_LDS_BANK_CONFLICT"][0],
        }
    out["_note"] = ("rocprofv3 --pmc passes (tools/pmc_passes.sh: separate runs for SQ / LDS / FETCH_SIZE / WRITE_SIZE+TCC, each with "
                    "--kernel-trace only) over `bench.py --steps 1 --warmup 1 --cpu-images 0 --no-secondary` (batch / precision as stamped below, n=16), mean per "
                    "dispatch.  hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE) KB: FETCH_SIZE is doubled per MI355X_MICROARCH.md §HBM "
                    "(gfx950 reports half the bytes of a wide coalesced read stream; Infinity-Cache hits are counted too); WRITE_SIZE "
                    "uncorrected.  mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 XCDs * 1024 SIMDs).")
    import bench
    out["kernel_sources_sha16"] = bench.kernel_sources_sha()
    out["batch"] = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    out["precision"] = sys.argv[4] if len(sys.argv) > 4 else "fp16"
    # the GPU box has no .git: the stamp is the UTC time of the measurement plus the commit handed over in MNET_GIT_COMMIT (or read from git here)
    import time
    commit = os.environ.get("MNET_GIT_COMMIT", "")
    if not commit:
        try:
            commit = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], text=True, stderr=subprocess.DEVNULL).strip()
        except Exception:      # noqa: BLE001
            commit = ""
    out["measured_at"] = time.strftime("%Y-%m-%d %H:%M UTC", time.gmtime()) + (" on commit " + commit if commit else "") + ", kernel sources " + out["kernel_sources_sha16"]
    json.dump(out, open(sys.argv[2], "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "_note"}, indent=1))


if __name__ == "__main__":
    main()
