#!/usr/bin/env python3
"""Transcribe the numeric known answers of the reference's own unit tests into JSON.

Build-container only (reads /root/reference as text).  Only numbers are kept:
  * 8 x 19 golden roots            (src/dynamic_interface.rs:268-503)
  * 9 outputs of the exp KAT       (src/linalg/storage.rs:498-520) + its 3x3 input
  * mode-id triples                (src/postprocessing.rs:424-498)
"""
import json
import os
import re

REF = "/root/reference/src"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_known_answers.json")
NUM = r"-?\d+\.\d*(?:[eE][-+]?\d+)?"


def main():
    ans = {}
    src = open(os.path.join(REF, "dynamic_interface.rs")).read()
    roots = {}
    for m in re.finditer(r"fn test_frequencies_(\w+)\(\)(.*?)\n    \}\n", src, re.S):
        name, body = m.group(1), m.group(2)
        arr = re.search(r"\[\s*((?:" + NUM + r"\s*,?\s*)+)\]", body).group(1)
        roots[name] = [float(x) for x in re.findall(NUM, arr)]
        assert len(roots[name]) == 19, (name, len(roots[name]))
    assert sorted(roots) == sorted(["colloc2", "colloc4", "colloc6", "colloc8",
                                    "magnus2", "magnus4", "magnus6", "magnus8"])
    ans["radial_roots_zams"] = {"model": "zams", "ell": 0, "m": 0, "grid": [1.0, 25.0, 25], "ulp": 1,
                                "roots": roots}
    src = open(os.path.join(REF, "linalg/storage.rs")).read()
    body = re.search(r"fn test_matrix_exponential\(\)(.*?)\n    \}\n", src, re.S).group(1)
    rows = re.search(r"from_row_slice\(&\[(.*?)\]\)", body, re.S).group(1)
    cols = re.search(r"from_column_slice\(&\[(.*?)\]\)", body, re.S).group(1)
    ans["exp_kat"] = {"input_row_major_3x3": [float(x) for x in re.findall(NUM, rows)],
                      "output_column_major_3x3": [float(x) for x in re.findall(NUM, cols)]}
    assert len(ans["exp_kat"]["input_row_major_3x3"]) == 9 and len(ans["exp_kat"]["output_column_major_3x3"]) == 9
    # mode id tests: (ell, m, lower, upper, steps, model) -> list of (cw, ccw, order)
    ans["mode_id"] = {
        "radial": {"args": [0, 0, 3.0, 25.0, 25, "zams"], "expected": [[0, i, i] for i in range(1, 20)]},
        "radial_other": {"args": [0, 0, 2.0, 25.0, 25, "joel"], "expected": [[0, i, i] for i in range(1, 18)]},
    }
    src = open(os.path.join(REF, "postprocessing.rs")).read()
    for name, args in (("dipole", [1, 0, 1.0, 25.0, 80, "zams"]), ("quadrupole", [2, 0, 2.0, 25.0, 80, "tams"])):
        body = re.search(r"fn test_mode_id_%s\(\)(.*?)\n    \}\n" % name, src, re.S).group(1)
        triples = re.findall(r"\((\d+),\s*(\d+),\s*(-?\d+)\)", body.split("vec![")[1])
        ans["mode_id"][name] = {"args": args, "expected": [[int(a), int(b), int(c)] for a, b, c in triples]}
    assert len(ans["mode_id"]["dipole"]["expected"]) == 21 and len(ans["mode_id"]["quadrupole"]["expected"]) == 22
    ans["polytrope0"] = {"gamma1": 5.0 / 3.0, "scheme": "colloc6", "npoints": 10000, "scan": [3.0, 4.0],
                         "n_roots": 1, "rel_tol": 1e-13}
    with open(OUT, "w") as f:
        json.dump(ans, f, indent=1)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
