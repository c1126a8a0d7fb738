"""print the headline fields of a bench.py line: python tools/experiments/print_line.py <file with the JSON line>"""
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); r = d["roofline"]
print("%s img/s %s ms | %s %s TFLOP/s frac %s | tail %s ms %s | traffic %s | all-levels %s no-image %s" % (
    d["value"], d["ms_per_step"], r["kernel"], r["achieved"], r["frac"], r["hbm_tail_ms_per_step"],
    {k: round(v["ms_per_step"], 1) for k, v in r["hbm_tail"]["by_kernel"].items()}, r.get("traffic"), d.get("value_all_levels_in_mode_precision"), d.get("value_without_prior_image")))
