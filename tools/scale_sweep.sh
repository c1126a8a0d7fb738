#!/bin/bash
# Scaling sweep for the day a multi-GPU node is there (nothing above N = 1 has been measured so far — DESIGN.md §7).
#     bash tools/scale_sweep.sh [outdir]        from the repo root on a node with up to 8 visible MI355X
# For N in {1, 2, 4, 8} (as many as are visible): BASELINE configs[2] (`bench.py --batch 128`, weak scaling: 128 strips per GPU, the
# uint8 all-gather of every step inside the timed region) and configs[4] (`--config mixed`, work-balanced shards, bucketed widths),
# then the gathered-vs-single-GPU bit-equality test (tests/test_multigpu_gpu.py, parametrised over 2 / 4 / 8 ranks).
# Output: one JSON line per run under <outdir>/ and a 4-row table per config: N, images/s, per-rank images/s, RCCL world size, scaling vs N = 1.
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
PY
done | tee "$OUT/table.txt"
timeout 3000 python -m pytest tests/test_multigpu_gpu.py -m gpu -q --tb=short 2>&1 | tail -5 | tee "$OUT/multigpu_equality.txt"
