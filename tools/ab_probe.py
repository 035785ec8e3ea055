"""Time the conv shape set under a few forced tile configs (tuning aid; MI355X only): python tools/ab_probe.py [cfg,sk ...]"""
import json
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from sweep_igemm import CHILD, ROOT, SHAPES  # noqa: E402


def main():
    names = [s[0] for s in SHAPES]
    forces = sys.argv[1:] or ["0,1", "1,1", "1,4", "auto"]
    print("force  " + "  ".join(f"{n:>14s}" for n in names))
    for force in forces:
        env = dict(os.environ)
        if force != "auto":
            env["SDM_IGEMM_FORCE"] = force
        r = subprocess.run([sys.executable, "-c", CHILD % (os.path.join(ROOT, "simple-diffusion-model_amd"), SHAPES)],
                           env=env, capture_output=True, text=True)
        try:
            v = json.loads(r.stdout.strip().splitlines()[-1])
            print(f"{force:6s} " + "  ".join(f"{v[n][0]:7.1f}us{v[n][1]:5.0f}T" for n in names), flush=True)
        except Exception:
            print(force, r.stderr[-400:], flush=True)


if __name__ == "__main__":
    main()
