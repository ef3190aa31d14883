#!/usr/bin/env python3
"""Re-encode the reference's quadrotor problem data as a JSON data file.

Run in the build container only (needs /root/reference).  The reference ships its
Crazyflie model as C++ headers of row-major literal arrays
(examples/problem_data/quadrotor_{20,50,100}hz_params.hpp: rho_value :5, Adyn :7, Bdyn :21,
Kinf :35, Pinf :41, Quu_inv :55, AmBKt :61, coeff_d2p :75, Q :89, R :91).  The numbers are
facts; we keep them as decimal strings exactly as printed there (so that a double->float
conversion reproduces what the reference's compiler does) in
accelerated-tinympc_amd/data/quadrotor_<rate>hz.json.  No source text is copied.
"""
import json, re, sys, pathlib

REF = pathlib.Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
OUT = pathlib.Path(__file__).resolve().parent.parent / "accelerated-tinympc_amd" / "data"

SHAPES = {"Adyn": ("nx", "nx"), "Bdyn": ("nx", "nu"), "Kinf": ("nu", "nx"), "Pinf": ("nx", "nx"),
          "Quu_inv": ("nu", "nu"), "AmBKt": ("nx", "nx"), "coeff_d2p": ("nx", "nu"), "Q": ("nx",), "R": ("nu",)}


def parse(path):
    txt = path.read_text()
    out = {"nx": 12, "nu": 4, "layout": "row-major", "source": str(path.relative_to(REF))}
    m = re.search(r"rho_value\s*=\s*([-0-9.eE+]+)\s*;", txt)
    out["rho"] = m.group(1)
    for name, shape in SHAPES.items():
        m = re.search(name + r"_data\s*\[[^\]]*\]\s*=\s*\{([^}]*)\}", txt, re.S)
        vals = [v.strip() for v in m.group(1).replace("\n", " ").split(",") if v.strip()]
        dims = [out[s] for s in shape]
        n = 1
        for d_ in dims:
            n *= d_
        assert len(vals) == n, (name, len(vals), n)
        out[name] = {"shape": dims, "values": vals}
    return out


def main():
    OUT.mkdir(parents=True, exist_ok=True)
    for rate in (20, 50, 100):
        p = REF / "examples" / "problem_data" / f"quadrotor_{rate}hz_params.hpp"
        d = parse(p)
        body = ",\n".join(f"{json.dumps(k)}:{json.dumps(v, separators=(',', ':'))}" for k, v in d.items())
        (OUT / f"quadrotor_{rate}hz.json").write_text("{\n" + body + "\n}\n")
        print("wrote", OUT / f"quadrotor_{rate}hz.json")


if __name__ == "__main__":
    main()
