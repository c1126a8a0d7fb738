#!/usr/bin/env python
"""profiles/pmc_traffic.json from the per-kernel counter table written by tools/pmc_summary.py (bench.py reads it to fill
roofline.traffic).  usage: python tools/pmc_traffic_json.py gpurun_out/pmc_bench.txt profiles/pmc_traffic.json <batch> <precision> [measured-at string]

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
    out, acc = {}, {}
    for b in re.split(r"\n(?=\S)", txt):
        lines = b.strip().split("\n")
        # template arguments: BC, BP, WC, WP, STAGES, MF, DBG = 0, X3, SPREAD, PIPE, MX, SWP, SGN (round 5: the software-pipelined tile built with the
        # GroupNorm-sum block, conv_dma_swp_gn.hip — the SAME tile id for bench.py, so its launches are folded into the "swp" entry)
        m = re.match(r"void conv_dma_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+), 0(?:, (\w+))?(?:, (\w+))?(?:, (\w+))?(?:, (\w+))?(?:, (\w+))?(?:, (\w+))?>", lines[0])
        w4 = re.match(r"void conv_dma_w4_kernel<", lines[0])        # round 6: the one-wave-per-SIMD tile (four builds by epilogue features: ONE tile id for bench.py)
        if not m and not w4:
            continue
        if m:
            x3, spread, pipe, mx, swp = (m.group(i) in ("true", "1") for i in (7, 8, 9, 10, 11))
        c = {}
        for l in lines[1:]:
            q = re.match(r"\s+(\S+)\s+mean/dispatch\s+([\d.]+)\s+dispatches (\d+)", l)
            if q:
                c[q.group(1)] = (float(q.group(2)), int(q.group(3)))
        key = "conv_dma_w4_kernel f16x2" if w4 else ("conv_dma_kernel<%s,mx%s> f16x2" % (",".join(m.groups()[:6]), ",swp" if swp else (",pipe" if pipe else "")) if mx else
               "conv_dma_kernel<%s%s>%s" % (",".join(m.groups()[:6]), (",spread" if spread else "") + (",pipe" if pipe else ""), " f16x3" if x3 else ""))
        a = acc.setdefault(key, {})
        for name, (mean, n) in c.items():        # totals over the dispatches of every build that carries this key
            t = a.setdefault(name, [0.0, 0])
            t[0] += mean * n
            t[1] += n
    for key, a in acc.items():
        mean = lambda name: a[name][0] / max(a[name][1], 1)      # noqa: E731
        f, w = mean("FETCH_SIZE"), mean("WRITE_SIZE")
        out[key] = {
            "hbm_bytes_per_launch": round((2 * f + w) * 1024),
            "FETCH_SIZE_KB_raw": f, "WRITE_SIZE_KB": w, "dispatches": a["FETCH_SIZE"][1],
            "l2_hit_rate": round(a["TCC_HIT_sum"][0] / (a["TCC_HIT_sum"][0] + a["TCC_MISS_sum"][0]), 4),
            "mfma_busy_frac_of_cycles": round(a["SQ_VALU_MFMA_BUSY_CYCLES"][0] / (a["GRBM_GUI_ACTIVE"][0] / 8.0 * 1024), 4),
            "lds_bank_conflict_cycles": mean("SQ_LDS_BANK_CONFLICT"),
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
    when = sys.argv[5] if len(sys.argv) > 5 else time.strftime("%Y-%m-%d %H:%M UTC", time.gmtime())     # (argv[5]: the measurement's own time when the JSON is rebuilt from a kept summary)
    out["measured_at"] = when + (" on commit " + commit if commit else "") + ", kernel sources " + out["kernel_sources_sha16"]
    json.dump(out, open(sys.argv[2], "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "_note"}, indent=1))


if __name__ == "__main__":
    main()
