"""A/B of 32- vs 64-deep K-steps of the 64x64 tile (SDM_IGEMM_BK64), repeated to expose outliers.  MI355X only."""
import json
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from sweep_igemm import CHILD, ROOT, SHAPES  # noqa: E402


def main():
    names = [s[0] for s in SHAPES]
    print("force bk64 " + "  ".join(f"{n:>14s}" for n in names))
    for rep in range(2):
        for force in ("1,1", "auto"):
            for bk in ("0", "2"):
                env = dict(os.environ, SDM_IGEMM_BK64=bk)
                if force != "auto":
                    env["SDM_IGEMM_FORCE"] = force
                r = subprocess.run([sys.executable, "-c", CHILD % (os.path.join(ROOT, "simple-diffusion-model_amd"), SHAPES)],
                                   env=env, capture_output=True, text=True)
                try:
                    v = json.loads(r.stdout.strip().splitlines()[-1])
                    print(f"{force:5s} {bk:4s} " + "  ".join(f"{v[n][0]:7.1f}us{v[n][1]:5.0f}T" for n in names), flush=True)
                except Exception:
                    print(force, bk, r.stderr[-400:], flush=True)


if __name__ == "__main__":
    main()
