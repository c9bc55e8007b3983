#!/usr/bin/env python3
"""Regenerates tests/golden/ from the reference.  Runs only in the build
container (needs /root/reference and oracle/_ref/libvsm_ref.so, see
oracle/Makefile); the outputs are committed so the GPU box needs neither.

  ex3_kd0.npy, ex3_kd1.npy  uint64[n,4]  descriptors of the reference's fixture
                            test/ex3_test_data/kd{0,1}.json (cereal JSON,
                            "m.corner_descriptors"[i].data = bitset::to_string(),
                            char 0 = bit 255), packed as bit k -> word k//64,
                            bit k%64.
  ex3_matches.npy           int32[220,2] test/ex3_test_data/matches.json
                            "m.matches" (the golden of test_ex3.cpp:148-185).
  ref_cases.json            per named case of tests/cases.py: the output of the
                            REAL reference function (count, sha256 of the int32
                            pair list, the list itself when short).
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF_DATA = "/root/reference/test/ex3_test_data"


def pack_bitstrings(entries):
    out = np.zeros((len(entries), 4), dtype=np.uint64)
    for i, e in enumerate(entries):
        assert e["type"] == 2, "only cereal bitset type::string entries are expected"
        s = e["data"]
        assert len(s) == 256
        v = int(s, 2)  # char 0 is bit 255
        for w in range(4):
            out[i, w] = (v >> (64 * w)) & 0xFFFFFFFFFFFFFFFF
    return out


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.int32).tobytes()).hexdigest()


def main():
    for name in ("kd0", "kd1"):
        with open(os.path.join(REF_DATA, name + ".json")) as f:
            kd = json.load(f)["value0"]
        np.save(os.path.join(HERE, f"ex3_{name}.npy"), pack_bitstrings(kd["m.corner_descriptors"]))
    with open(os.path.join(REF_DATA, "matches.json")) as f:
        m = json.load(f)["value0"]["m.matches"]
    np.save(os.path.join(HERE, "ex3_matches.npy"),
            np.array([[e["first"], e["second"]] for e in m], dtype=np.int32))

    from oracle import binding
    from tests import cases
    ref = binding.ref()
    assert ref is not None, "build oracle/_ref first: make -C oracle ref"

    out = {"_generator": "tests/golden/make_golden.py",
           "_source": "real reference function, /root/reference/include/visnav/keypoints.h:247-361",
           "cases": {}}
    for name in cases.case_names():
        d1, d2, thr, ratio = cases.build_case(name)
        res = ref.match(d1, d2, thr, ratio)
        rec = {"n1": int(len(d1)), "n2": int(len(d2)), "threshold": thr,
               "dist_2_best": repr(float(ratio)), "count": int(len(res)), "sha256": sha(res)}
        if len(res) <= 48:
            rec["pairs"] = res.tolist()
        out["cases"][name] = rec
        print(f"{name:34s} {rec['n1']:5d} x {rec['n2']:5d} thr {thr:4d} ratio {ratio!r:10} -> {rec['count']}")

    db, off, pairs = cases.sfm_small()
    recs = []
    for (i, j) in pairs.tolist():
        res = ref.match(db[off[i]:off[i + 1]], db[off[j]:off[j + 1]], 70, 1.2)
        recs.append({"pair": [i, j], "count": int(len(res)), "sha256": sha(res)})
    out["sfm_small"] = {"spec": cases.SFM_SMALL, "threshold": 70, "dist_2_best": "1.2",
                        "pairs": recs}
    print("sfm_small:", [r["count"] for r in recs])

    with open(os.path.join(HERE, "ref_cases.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
