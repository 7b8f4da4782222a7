"""GPU test of tools/storm_batch (the batch-input pipeline of the reference's rotation sweeps, reference
scripts/generate-batch-input-deformed.py and src/main.rs:560-640, executed in lock-step groups of blocks) against
the CPU oracle running the same command list line by line: every bracket, root, mode id and all six coupling
matrices of every block, bit for bit.  The reference asserts none of these values ("parity unpinned": GPU == oracle);
what it pins on this path (golden roots, mode ids) is covered in test_oracle_golden.py / test_hip_postprocess.py."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "storm_batch.bin")

BATCH = """\
input test-data/test-model-zams.GSM
# three rotation rates; the generator's positional precision and the CLI's --precision both occur
set-rotation-constant 0.02
scan 1 1 2.0 12.0 60 1e-8
scan 3 1 2.0 12.0 50 --precision 1e-9
scan 2 0 2.0 9.0 40 1e-8 --inverse
scan 1 -1 2.0 9.0 30 1e-8 --difference-scheme colloc2
post-process
deform 0.02
perturb-deformed 1
perturb-deformed 0
output rate-a
set-rotation-constant 0.05
scan 1 1 2.0 12.0 60 1e-8
scan 3 1 2.0 12.0 50 1e-8
post-process
deform 0.05
perturb-deformed 1
output rate-b
scan 0 0 3.0 14.0 40 1e-8
scan 2 0 3.0 14.0 40 1e-8
post-process
perturb-deformed 0
output rate-c
"""


def hexes(line):
    return [float.fromhex(t) for t in line]


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def parse_output(path):
    blocks = []
    with open(path) as f:
        for raw in f:
            w = raw.split()
            if w[0] == "block":
                blocks.append({"label": w[2], "scans": [], "coupling": []})
            elif w[0] == "scan":
                blocks[-1]["scans"].append({"ell": int(w[1]), "m": int(w[2]), "modes": []})
            elif w[0] == "mode":
                blocks[-1]["scans"][-1]["modes"].append((hexes(w[1:6]), [int(t) for t in w[6:]]))
            elif w[0] == "coupling":
                blocks[-1]["coupling"].append({"m": int(w[1]), "n": int(w[3]), "mats": []})
            elif w[0] == "matrix":
                blocks[-1]["coupling"][-1]["mats"].append(np.array(hexes(w[2:])))
    return blocks


# every scan with the same scheme and precision: each group is ONE sweep, the fields stay in HBM and the integrals of all
# blocks of a group (different sizes, different sets of degrees, two azimuthal orders) go out as one launch
UNIFORM = """\
input test-data/test-model-zams.GSM
set-rotation-constant 0.03
scan 1 1 2.0 12.0 60 1e-8
scan 2 0 2.0 9.0 40 1e-8 --inverse
scan 3 1 2.0 12.0 50 1e-8
post-process
deform 0.03
perturb-deformed 1
perturb-deformed 0
output u-a
set-rotation-constant 0.06
scan 2 1 2.0 10.0 45 1e-8
scan 4 1 2.0 10.0 45 1e-8
post-process
deform 0.06
perturb-deformed 1
output u-b
set-rotation-constant 0.09
scan 0 0 3.0 14.0 40 1e-8
scan 2 0 3.0 14.0 40 1e-8
scan 1 -1 3.0 9.0 30 1e-8
post-process
deform 0.09
perturb-deformed 0
perturb-deformed -1
output u-c
"""


def oracle_run(oracle, golden_dir, text=None):
    """The command list of BATCH executed one line at a time with the CPU oracle (StormState semantics)."""
    model = oracle.Model.from_gsm_npz(os.path.join(golden_dir, "gsm_zams.npz"))
    grid = model.field("r_coord")
    blocks, cur, metric = [], {"scans": [], "coupling": []}, None
    post = None
    for raw in (BATCH if text is None else text).splitlines():
        w = raw.split()
        if not w or w[0].startswith("#") or w[0] == "input":
            continue
        if w[0] == "set-rotation-constant":
            model.set_rotation_constant(float(w[1]))
        elif w[0] == "scan":
            pos, inverse, prec, scheme = [], False, 1e-8, "colloc4"
            it = iter(w[1:])
            for t in it:
                if t == "--inverse":
                    inverse = True
                elif t == "--precision":
                    prec = float(next(it))
                elif t == "--difference-scheme":
                    scheme = next(it)
                else:
                    pos.append(t)
            if len(pos) == 6:
                prec = float(pos[5])
            ell, m, lo, hi, steps = int(pos[0]), int(pos[1]), float(pos[2]), float(pos[3]), int(pos[4])
            o = oracle.Solver(model, ell, m, scheme, grid)
            f = oracle.rev_linspace(lo, hi, steps) if inverse else oracle.linspace(lo, hi, steps)
            res, _, _ = o.scan_and_optimize(f, oracle.relative_to_ulp(prec), threads=16)
            roots = np.array([b["root"] for b in res])
            _, vecs = o.eigenvector(roots, threads=16) if roots.size else (None, np.empty((0, grid.size * 4)))
            cur["scans"].append({"ell": ell, "m": m, "res": res, "roots": roots, "vecs": vecs})
        elif w[0] == "post-process":
            for s in cur["scans"]:
                s["post"] = model.postprocess(s["ell"], s["m"], s["roots"], s["vecs"], threads=16) if s["roots"].size else None
        elif w[0] == "deform":
            metric = model.perturb_structure(float(w[1]))
        elif w[0] == "perturb-deformed":
            m = int(w[1])
            sel = [s for s in cur["scans"] if s["m"] == m and s["roots"].size]
            ells = np.concatenate([np.full(s["roots"].size, s["ell"], dtype=np.uint64) for s in sel])
            freqs = np.concatenate([s["roots"] for s in sel])
            fields = np.concatenate([s["post"]["fields"] for s in sel])
            mats, _ = model.coupling_matrices(ells, freqs, fields, m, metric, threads=16)
            cur["coupling"].append({"m": m, "n": freqs.size, "mats": mats})
        elif w[0] == "output":
            cur["label"] = w[1]
            blocks.append(cur)
            cur = {"scans": [], "coupling": []}
    return blocks


@pytest.mark.skipif(not os.path.exists(TOOL), reason="tools/storm_batch.bin not built (make -C storm_amd/csrc)")
@pytest.mark.parametrize("text,labels", [(BATCH, ["rate-a", "rate-b", "rate-c"]), (UNIFORM, ["u-a", "u-b", "u-c"])])
def test_batch_pipeline_matches_line_by_line_oracle(oracle, golden_dir, tmp_path, text, labels):
    from storm_amd import ffi
    model_bin = tmp_path / "zams_model.bin"
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_model_bin.py"), "zams", str(model_bin)], check=True)
    batch = tmp_path / "input"
    batch.write_text(text)
    outs = []
    for group in (1, 2, 16):
        out = tmp_path / f"out{group}.txt"
        r = subprocess.run([TOOL, str(model_bin), str(batch), str(out), "--group", str(group)], capture_output=True, text=True,
                           timeout=300)
        assert r.returncode == 0, r.stderr
        outs.append(out.read_text())
    assert outs[0] == outs[1] == outs[2]     # grouping blocks changes nothing
    # blocks are independent: three processes with one share each write the three parts of the same record
    parts = []
    for k in range(3):
        out = tmp_path / f"shard{k}.txt"
        r = subprocess.run([TOOL, str(model_bin), str(batch), str(out), "--shard", f"{k}/3", "--device", "0"], capture_output=True,
                           text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        parts.append(out.read_text())
    renumber = lambda t: "\n".join(" ".join(["block", "*"] + ln.split()[2:]) if ln.startswith("block ") else ln for ln in t.splitlines())
    assert renumber("".join(parts)) == renumber(outs[0])
    got = parse_output(tmp_path / "out2.txt")
    ref = oracle_run(oracle, golden_dir, text)
    assert [b["label"] for b in got] == [b["label"] for b in ref] == labels
    total = 0
    for gb, rb in zip(got, ref):
        assert len(gb["scans"]) == len(rb["scans"])
        for gs, rs in zip(gb["scans"], rb["scans"]):
            assert (gs["ell"], gs["m"]) == (rs["ell"], rs["m"]) and len(gs["modes"]) == len(rs["res"])
            for k, ((vals, ints), b) in enumerate(zip(gs["modes"], rs["res"])):
                want = [b["lo_x"], b["lo_f"], b["hi_x"], b["hi_f"], b["root"]]
                assert np.array_equal(bits(vals), bits(want)), (gb["label"], gs["ell"], k)
                assert ints[0] == b["evals"]
                if rs.get("post") is not None:
                    assert ints[1:4] == rs["post"]["ids"][k].tolist() and ints[4] == int(rs["post"]["n_nodes"][k])
                total += 1
        assert len(gb["coupling"]) == len(rb["coupling"])
        for gc, rc in zip(gb["coupling"], rb["coupling"]):
            assert (gc["m"], gc["n"]) == (rc["m"], rc["n"]) and rc["n"] > 0
            for q, name in enumerate(ffi.COUPLING_MATRICES):
                assert np.array_equal(bits(gc["mats"][q]), bits(rc["mats"][name].T.ravel())), (gb["label"], gc["m"], name)
    assert total > 30


@pytest.mark.skipif(not os.path.exists(TOOL), reason="tools/storm_batch.bin not built (make -C storm_amd/csrc)")
def test_batch_rejects_what_the_reference_rejects(tmp_path):
    model_bin = tmp_path / "zams_model.bin"
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_model_bin.py"), "zams", str(model_bin)], check=True)
    for text, msg in (("scan 1 0 1 2 10\nperturb-deformed 0\n", "Deformed structure was not computed"),
                      ("deform 0.1\nscan 1 0 1 2 10\nperturb-deformed 0\n", "Post processing was not ran"),
                      ("frobnicate\n", "unsupported command"),
                      ("scan 1 0 1 2 10 --frequency-units hertz\n", "only dynamical units")):
        batch = tmp_path / "bad"
        batch.write_text(text)
        r = subprocess.run([TOOL, str(model_bin), str(batch), str(tmp_path / "o.txt")], capture_output=True, text=True, timeout=60)
        assert r.returncode == 2 and msg in r.stderr, (text, r.stderr)
