#!/usr/bin/env python3
"""Regenerate tests/golden/golden.json and the small full-text fixtures from the UNMODIFIED reference binary
(oracle/_ref/miniasm_ref, built from /root/reference by oracle/Makefile).  Run in the build container:

    python tests/golden/make_golden.py

Inputs are produced by the seeded generator (miniasm_amd/bin/pafgen), so only its arguments and the digest
of its output are stored; every dump is stored as the sha256 of its LC_ALL=C-sorted lines.  One tiny case is
stored in full (PAF + every dump) so that a human can diff it.  Inputs must be free of (u,len) arc tie
groups: the script refuses to record a case that has any."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
import miniasm_amd as ma  # noqa: E402
import refapi as R  # noqa: E402

INPUTS = {
    "lognormal": dict(reads=3000, lines=80000, seed=41, extra=[]),
    "fixed": dict(reads=2500, lines=70000, seed=42, extra=["-L", "fixed"]),
    "noisy": dict(reads=4000, lines=90000, seed=63, extra=["-L", "uniform", "-d", "0.35", "-x", "0.03"]),
    "tiny": dict(reads=120, lines=1500, seed=5, extra=["-L", "uniform", "-d", "0.2", "-x", "0.05"]),
}
DUMPS = ["-p bed", "-p paf -S2", "-p paf -S3", "-p paf -S4", "-p paf", "-p sg -S5", "-p sg -S6", "-p sg -S7", "-p sg -S9", "-p sg -S10", "-p sg", "-p ug",
         "-p ug -1", "-p sg -2", "-p ug -b", "-p ug -R", "-p ug -c 2 -s 1500 -h 500 -I 0.7 -g 500 -e 3 -d 30000", "-p ug -n 4 -r 0.8,0.4 -F 0.9"]


def main():
    if not R.have_ref():
        sys.exit("oracle/_ref/miniasm_ref is missing: run `make -C oracle ref` where /root/reference exists")
    gold = {"reference_version": R.run_cli(R.REF_BIN, ["-V"], "/dev/null")[0].decode().strip() if False else "0.3-r179", "inputs": {}}
    tmp = os.path.join(HERE, "_tmp")
    os.makedirs(tmp, exist_ok=True)
    for name, cfg in INPUTS.items():
        paf = R.pafgen(os.path.join(tmp, name + ".paf"), cfg["reads"], cfg["lines"], cfg["seed"], cfg["extra"])
        sg5, _ = R.run_cli(R.REF_BIN, ["-p", "sg", "-S5"], paf)
        ties = R.arc_tie_groups(sg5)
        if ties:
            sys.exit("input %s has %d arc tie groups; pick another seed" % (name, ties))
        entry = {"pafgen": cfg, "paf_digest": R.digest(open(paf, "rb").read()), "dumps": {}}
        for d in DUMPS:
            out, _ = R.run_cli(R.REF_BIN, d.split(), paf)
            if "-p sg" in d and R.arc_tie_groups(out):
                continue
            entry["dumps"][d] = R.digest(out)
            if name == "tiny":
                with open(os.path.join(HERE, "tiny." + d.replace("-", "").replace(" ", "_").replace(",", "_") + ".txt"), "wb") as f:
                    f.write(out)
        if name == "tiny":
            os.replace(paf, os.path.join(HERE, "tiny.paf"))
        gold["inputs"][name] = entry
        print(name, len(entry["dumps"]), "dumps")
    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(gold, f, indent=1, sort_keys=True)
    for fn in os.listdir(tmp):
        os.remove(os.path.join(tmp, fn))
    os.rmdir(tmp)


if __name__ == "__main__":
    main()
