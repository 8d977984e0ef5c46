"""Reader for the reference's JSON problem format (default/json.rs:13-21, serde encoding of
SupportedConeT in cones/supportedcone.rs): returns the dict layout of tests/e2e_problems.py."""
import json

import numpy as np

TAGS = {"ZeroConeT": 0, "NonnegativeConeT": 1, "SecondOrderConeT": 2, "ExponentialConeT": 3, "PowerConeT": 4,
        "GenPowerConeT": 5, "PSDTriangleConeT": 6}


def _csc(d):
    return (np.asarray(d["colptr"], dtype=np.int64), np.asarray(d["rowval"], dtype=np.int64),
            np.asarray(d["nzval"], dtype=np.float64))


def _cone(c):
    if isinstance(c, str):  # unit variant: "ExponentialConeT"
        return (TAGS[c], 3)
    (name, val), = c.items()
    if name == "PowerConeT":
        return (TAGS[name], 3, 0, float(val))
    if name == "GenPowerConeT":
        raise NotImplementedError("GenPowerConeT fixtures")
    return (TAGS[name], int(val))


def load(path):
    d = json.load(open(path))
    assert d["P"]["m"] == d["P"]["n"] == d["A"]["n"] and d["A"]["m"] == len(d["b"])
    P = _csc(d["P"])
    for col in range(d["P"]["n"]):  # the path takes P as its upper triangle (data.P = P.to_triu())
        assert all(r <= col for r in P[1][P[0][col]:P[0][col + 1]]), "P must be upper triangular"
    return dict(n=d["P"]["n"], m=d["A"]["m"], P=P, A=_csc(d["A"]), q=list(d["q"]), b=list(d["b"]),
                cones=[_cone(c) for c in d["cones"]], settings=d.get("settings", {}))
