#!/bin/bash
# Scaling sweep for the day a multi-GPU node is there (nothing above N = 1 has been measured so far — DESIGN.md §7).
#     bash tools/scale_sweep.sh [outdir]        from the repo root on a node with up to 8 visible MI355X
# For N in {1, 2, 4, 8} (as many as are visible): BASELINE configs[2] (`bench.py --batch 128`, weak scaling: 128 strips per GPU, the
# uint8 all-gather of every step inside the timed region) and configs[4] (`--config mixed`, work-balanced shards, bucketed widths),
# then the gathered-vs-single-GPU bit-equality test (tests/test_multigpu_gpu.py, parametrised over 2 / 4 / 8 ranks).
# Output: one JSON line per run under <outdir>/, a 4-row table per config (N, images/s, per-rank images/s, RCCL world size, ratio to N x the
# N = 1 rate) and <outdir>/SCALE_manual.json — ONE record of the same content (per config: the rows; the equality test's result; GPUs visible;
# commit; UTC time), shaped so that it can stand where the driver's SCALE_rNN.json stands the day a node exists, without edits.
set -uo pipefail
OUT="${1:-gpurun_out/scale_sweep}"; mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
NGPU=$(python -c 'import torch; print(torch.cuda.device_count())')
echo "[scale_sweep] $NGPU GPU(s) visible"
for cfg in "sr:--batch 128" "mixed:--config mixed --batch 128"; do
  name=${cfg%%:*}; args=${cfg#*:}
  for n in 1 2 4 8; do
    [ "$n" -le "$NGPU" ] || continue
    log="$OUT/${name}_n${n}.json"
    if [ "$n" -eq 1 ]; then
      timeout 1800 python bench.py --gpus 1 --steps 5 --warmup 2 --no-secondary --cpu-images 0 $args 2>"$OUT/${name}_n${n}.err" | tail -1 > "$log"
    else
      timeout 1800 python -m torch.distributed.run --nnodes=1 --nproc-per-node "$n" --master-addr 127.0.0.1 --master-port $((29600 + n)) \
        bench.py --gpus "$n" --steps 5 --warmup 2 --no-secondary --cpu-images 0 $args 2>"$OUT/${name}_n${n}.err" | tail -1 > "$log"
    fi
  done
  python - "$OUT" "$name" <<'PY'
import json, os, sys
out, name = sys.argv[1], sys.argv[2]
rows, base = [], None
for n in (1, 2, 4, 8):
    p = os.path.join(out, "%s_n%d.json" % (name, n))
    if not os.path.isfile(p):
        continue
    try:
        d = json.loads(open(p).read())
    except Exception as e:      # noqa: BLE001
        print("%s N=%d: no JSON line (%s) — see %s" % (name, n, e, p.replace(".json", ".err")))
        continue
    base = base or d["value"] / max(d["n_gpus"], 1)
    r = d.get("ranks", {})
    rows.append((n, d["value"], r.get("per_rank_images_per_s", [d["value"]]), r.get("world_size", 1), d["value"] / (base * n)))
print("config %-6s  N   images/s   per-rank images/s                         RCCL world   vs N x (N=1)" % name)
for n, v, pr, w, eff in rows:
    print("               %d   %8.1f   %-42s %-10s   %.3f" % (n, v, " ".join("%.1f" % x for x in pr), w, eff))
json.dump([{"n_gpus": n, "value": v, "unit": "images/s", "per_rank_images_per_s": pr, "rccl_world_size": w, "ratio_to_N_times_N1": round(eff, 4)}
           for n, v, pr, w, eff in rows], open(os.path.join(out, "rows_%s.json" % name), "w"))
PY
done | tee "$OUT/table.txt"
timeout 3000 python -m pytest tests/test_multigpu_gpu.py -m gpu -q --tb=short 2>&1 | tail -5 | tee "$OUT/multigpu_equality.txt"
python - "$OUT" "$NGPU" <<'PY'
import json, os, subprocess, sys, time
out, ngpu = sys.argv[1], int(sys.argv[2])
rec = {"skipped": False, "source": "tools/scale_sweep.sh (builder- or operator-run; the driver's own SCALE record supersedes it)",
       "gpus_visible": ngpu, "measured_at_utc": time.strftime("%Y-%m-%d %H:%M", time.gmtime()), "scaling": "weak",
       "commit": subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None,
       "target": "north_star: >= 2000 SR images/s on 8 x MI355X (8 x the single-GPU rate of BENCH is the projection; this file is the measurement)",
       "configs": {}}
for name, what in (("sr", "BASELINE configs[2]: batch 128 per GPU, uint8 all-gather of every step inside the timed region"),
                   ("mixed", "BASELINE configs[4]: mixed widths, work-balanced shards")):
    p = os.path.join(out, "rows_%s.json" % name)
    rec["configs"][name] = {"workload": what, "rows": json.load(open(p)) if os.path.isfile(p) else []}
eq = open(os.path.join(out, "multigpu_equality.txt")).read().strip().splitlines()
rec["gathered_equals_single_gpu_test"] = eq[-1] if eq else "not run"
if ngpu < 2:
    rec["note"] = "fewer than 2 GPUs visible: only the N = 1 rows exist; the equality tests for 2 / 4 / 8 ranks were skipped"
json.dump(rec, open(os.path.join(out, "SCALE_manual.json"), "w"), indent=1)
print("[scale_sweep] wrote %s" % os.path.join(out, "SCALE_manual.json"))
PY
