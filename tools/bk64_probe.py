"""Compare 32-deep and 64-deep K-steps of the small-tile contraction kernel (tuning aid; MI355X only)."""
import json
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from sweep_igemm import CHILD, ROOT, SHAPES  # noqa: E402


def main():
    names = [s[0] for s in SHAPES]
    print("force  bk64  " + "  ".join(f"{n:>14s}" for n in names))
    for force in ("1,1", "1,2", "1,4", "2,1", ""):
        for bk in ("0", "2", "3"):
            env = dict(os.environ, SDM_IGEMM_BK64=bk)
            if force:
                env["SDM_IGEMM_FORCE"] = force
            r = subprocess.run([sys.executable, "-c", CHILD % (os.path.join(ROOT, "simple-diffusion-model_amd"), SHAPES)],
                               env=env, capture_output=True, text=True)
            try:
                v = json.loads(r.stdout.strip().splitlines()[-1])
                print(f"{force or 'auto':6s} {bk:4s}  " + "  ".join(f"{v[n][0]:7.1f}us{v[n][1]:5.0f}T" for n in names), flush=True)
            except Exception:
                print(force, bk, r.stderr[-400:], flush=True)


if __name__ == "__main__":
    main()
