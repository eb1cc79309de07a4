"""BASELINE.json configs[4] at its FULL size on one GPU: 100 M power-law boxes are far beyond what the oracle can follow, so the
check is by size-independent properties (tools/check_large.py): sorted tree, strictly increasing pair list, no self pairs,
idempotent scan -- and a checksum of the pair list that a SECOND pipeline must reproduce: the same command with the expanding
scan, the grouped finishes and the segment finishes switched off runs the per-entry ancestor walk and full-width sorts, i.e.
other kernels end to end."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _check(n, env_extra):
    env = dict(os.environ)
    env.update(env_extra)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_large.py"), str(n), "powerlaw"], env=env, cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:]
    m = re.search(r"CHECKSUM objects (\d+) powerlaw pairs (\d+) sum (\d+)", r.stdout)
    assert m, r.stdout[-2000:]
    props = re.search(r"tree sorted (\w+), pairs strictly increasing (\w+), no self pairs (\w+), idempotent (\w+)", r.stdout)
    assert props and props.groups() == ("True", "True", "True", "True"), r.stdout[-2000:]
    return int(m.group(2)), int(m.group(3))


def test_hundred_million_powerlaw_properties_and_cross_pipeline_checksum():
    n = int(os.environ.get("BP_LARGE_OBJECTS", "100000000"))
    fast = _check(n, {})
    other = _check(n, {"BP_NO_EXPAND_SCAN": "1", "BP_NO_GROUP_FINISH": "1", "BP_NO_SEGMENT_FINISH": "1", "BP_RADIX_CLAIM": "0"})
    assert fast[0] > n and fast == other, (fast, other)
