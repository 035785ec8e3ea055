"""A/B of the XCD-aware tile order of the contraction kernel (tuning aid; MI355X only)."""
import json
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from sweep_igemm import CHILD, ROOT, SHAPES  # noqa: E402


def main():
    names = [s[0] for s in SHAPES]
    print("order  " + "  ".join(f"{n:>14s}" for n in names))
    for order in ("0", "n", "m", "auto", "0", "auto"):
        env = dict(os.environ)
        if order != "auto":
            env["SDM_IGEMM_XCD_ORDER"] = order
        r = subprocess.run([sys.executable, "-c", CHILD % (os.path.join(ROOT, "simple-diffusion-model_amd"), SHAPES)],
                           env=env, capture_output=True, text=True)
        try:
            v = json.loads(r.stdout.strip().splitlines()[-1])
            print(f"{order:6s} " + "  ".join(f"{v[n][0]:7.1f}us{v[n][1]:5.0f}T" for n in names), flush=True)
        except Exception:
            print(order, r.stderr[-400:], flush=True)


if __name__ == "__main__":
    main()
