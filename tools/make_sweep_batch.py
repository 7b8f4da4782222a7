"""A batch input of the shape scripts/generate-batch-input-deformed.py writes (reference), in dynamical units:
per rotation rate  set-rotation-constant / 8 scans (ell = 1, 3, 5, 7: 300 frequencies linear in 1..30 and 100 linear
in period in 0.5..1) / post-process / deform / perturb-deformed / output.
usage: python tools/make_sweep_batch.py out.txt [rates=64] [m=1] [max_rot=0.2]"""
import sys

out = sys.argv[1]
rates = int(sys.argv[2]) if len(sys.argv) > 2 else 64
m = int(sys.argv[3]) if len(sys.argv) > 3 else 1
max_rot = float(sys.argv[4]) if len(sys.argv) > 4 else 0.2
lines = ["input test-data/test-model-tams.GSM\n"]
for i in range(rates):
    rot = (i + 1) / rates * max_rot
    lines.append(f"set-rotation-constant {rot!r}\n")
    for ell in (1, 3, 5, 7):
        lines.append(f"scan {ell} {m} 1.0 30 300 1e-6\n")
        lines.append(f"scan {ell} {m} 0.5 1. 100 1e-6 --inverse\n")
    lines += ["post-process\n", f"deform {rot!r}\n", f"perturb-deformed {m}\n", f"output rate-{i}\n"]
with open(out, "w") as f:
    f.writelines(lines)
