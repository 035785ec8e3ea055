"""Split-K depth sweep on the weight-streaming layers (M = 64 and 256 rows at batch 4).  MI355X only."""
import json
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from sweep_igemm import CHILD, ROOT  # noqa: E402

SHAPES = [("c512@4", 4, 4, 4, 512, 512, "c3s1"), ("c1024@4", 4, 4, 4, 1024, 1024, "c3s1"), ("c512@8", 4, 8, 8, 512, 512, "c3s1"),
          ("c1024@8", 4, 8, 8, 1024, 1024, "c3s1"), ("ct1024-512@4", 4, 4, 4, 1024, 512, "ct4"), ("c512-256s2@32", 4, 32, 32, 512, 256, "c3s2")]


def main():
    names = [s[0] for s in SHAPES]
    print("force   " + "  ".join(f"{n:>14s}" for n in names))
    for force in ("auto", "1,8", "1,16", "1,24", "1,36", "1,48", "1,72", "2,16", "2,36", "2,72"):
        env = dict(os.environ)
        if force != "auto":
            env["SDM_IGEMM_FORCE"] = force
        r = subprocess.run([sys.executable, "-c", CHILD % (os.path.join(ROOT, "simple-diffusion-model_amd"), SHAPES)],
                           env=env, capture_output=True, text=True)
        try:
            v = json.loads(r.stdout.strip().splitlines()[-1])
            print(f"{force:7s} " + "  ".join(f"{v[n][0]:7.1f}us{v[n][1]:5.0f}T" for n in names), flush=True)
        except Exception:
            print(force, r.stderr[-400:], flush=True)


if __name__ == "__main__":
    main()
