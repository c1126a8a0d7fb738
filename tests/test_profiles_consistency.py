"""The bench line's `roofline.traffic` comes from a committed PMC record (profiles/pmc_traffic_fp16x2.json) and is reported only while the
record is keyed to the kernel sources in the tree (bench.kernel_sources_sha).  These CPU tests keep the two from drifting apart silently:
an edit of the dominant kernel's sources without new PMC passes — or a record whose kernel names the bench no longer produces — fails here
instead of turning `traffic` into null at the end of a round."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_for_profiles_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_pmc_record_is_keyed_to_the_sources_in_the_tree():
    bench = _bench()
    rec = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic_fp16x2.json")))
    assert rec["kernel_sources_sha16"] == bench.kernel_sources_sha(), (
        "conv_igemm_dma.hip / conv_dma_w4.hip / conv_dma_common.h / conv_args.h changed after the PMC passes: re-run tools/pmc_passes.sh + tools/pmc_traffic_json.py")
    assert rec["batch"] == 256 and rec["precision"] == "fp16x2"          # the default bench.py command's workload


def test_pmc_record_names_the_dominant_kernel_the_bench_reports():
    bench = _bench()
    rec = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic_fp16x2.json")))
    # the name bench.py builds for the dominant tile of the default mode (kname + dtype suffix): since round 6 the one-wave-per-SIMD 256x256 MX tile (fp16+8 id 16;
    # its four feature builds are ONE tile id for bench.py, merged by tools/pmc_traffic_json.py)
    tile = "conv_dma_w4_kernel"
    kid = [k for k, v in bench.KNAME_X2.items() if v == tile]
    assert len(kid) == 1
    name = bench.kname(kid[0], bench.PDT["fp16x2"]) + " " + bench.DTNAME[bench.PDT["fp16x2"]]
    assert name in rec, sorted(k for k in rec if k.startswith("conv"))
    ent = rec[name]
    alg = 11.7e9                                                         # algorithmic bytes per launch of that tile
    assert alg < ent["hbm_bytes_per_launch"] < 3 * alg
    assert 0.3 < ent["mfma_busy_frac_of_cycles"] < 1.0 and 0.5 < ent["l2_hit_rate"] <= 1.0
    assert ent["dispatches"] % 56 == 0                                   # 56 launches of the tile per step, all builds of it merged
    assert bench.DTNAME[3] == "f16x2"
