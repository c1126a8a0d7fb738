"""-m gpu: kernels that exist in two forms behind an A/B environment knob give the same BYTES in both (round 5: the ToRGB kernel with four
trips per workgroup), and the image-only generator level's f16 conversion riding in the up-sample's store leaves the SR output's bytes unchanged.  The knobs are read once per process: one worker process per setting."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(env_extra, *args):
    env = dict(os.environ)
    env.update(env_extra)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "ab_forms_worker.py")] + list(args), env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("AB_DIGESTS ")]
    assert line, r.stdout[-1000:]
    return json.loads(line[-1][len("AB_DIGESTS "):])


def test_torgb_forms_are_bit_identical():
    new = _run({})
    old = _run({"MNET_TORGB_TRIPS": "1"})
    assert new.keys() == old.keys() and len(new) == 8
    for k in new:
        assert new[k] == old[k], k


def test_image_level_conversion_in_the_upsample_leaves_sr_bytes_unchanged():
    new = _run({}, "--chain")
    old = _run({"MNET_NO_FUSE_IMG_CONVERT": "1"}, "--chain")
    assert new["chain.sr.fp16x2"] == old["chain.sr.fp16x2"]
