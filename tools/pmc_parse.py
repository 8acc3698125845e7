#!/usr/bin/env python3
"""Turn the two rocprofv3 --pmc passes of tools/pmc_run.py into a per-launch HBM traffic figure.

Units/corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE
are in KiB (x1024 -> bytes); on gfx950 FETCH_SIZE under-reports wide coalesced streams, so each counter
is scaled by the factor measured on the calibration copies of known size in the same access shape."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

from kbench import PRESETS


def per_kernel(outdir, counter):
    rows = defaultdict(list)
    files = glob.glob(os.path.join(outdir, "**", "*counter_collection.csv"), recursive=True)
    assert files, f"no counter_collection.csv under {outdir}"
    for f in files:
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == counter:
                rows[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return rows


def main():
    out, spec = sys.argv[1], sys.argv[2]
    N, E, G, delta = PRESETS[spec]
    fetch = per_kernel(os.path.join(out, "fetch"), "FETCH_SIZE")
    write = per_kernel(os.path.join(out, "write"), "WRITE_SIZE")
    res = {"workload": spec, "N": N, "E": E, "unit_note": "counter values in KiB; bytes = value * 1024"}

    def pick(d, key):
        ks = [k for k in d if key in k]
        assert ks, (key, list(d)[:5])
        return d[max(ks, key=lambda k: len(d[k]))]      # the variant launched most often (= the step kernel)

    # calibration: launches are ordered (20 MiB x3 @8B, x3 @16B, 1 GiB x3 @8B, x3 @16B)
    cal = {}
    for name, key in (("b64", "calib_copy_b64"), ("b128", "calib_copy_b128")):
        fv, wv = pick(fetch, key), pick(write, key)
        small, big = 20 * 2 ** 20, 2 ** 30
        cal[name] = {"fetch_ratio_20MiB": sum(fv[0:3]) / 3 * 1024 / small, "fetch_ratio_1GiB": sum(fv[3:6]) / 3 * 1024 / big,
                     "write_ratio_20MiB": sum(wv[0:3]) / 3 * 1024 / small, "write_ratio_1GiB": sum(wv[3:6]) / 3 * 1024 / big}
    wz = pick(write, "calib_write_z")
    cal["strided_z_nbr"] = {"write_ratio": sum(wz) / len(wz) * 1024 / (64 * 4096 * 36)}
    res["calibration_counter_over_true_bytes"] = cal
    fk, wk = pick(fetch, "drone_kernel"), pick(write, "drone_kernel")
    raw_f = sum(fk) / len(fk) * 1024
    raw_w = sum(wk) / len(wk) * 1024
    cf = cal["b64"]["fetch_ratio_20MiB"]; cw = cal["b64"]["write_ratio_20MiB"]
    layer = not os.environ.get("PLAIN")            # episode layer: + 32 B read and 32 B written per env (the record)
    alg_read, alg_write = 16 * N * E + (4 + (32 if layer else 0)) * E, 60 * N * E + (9 + (32 if layer else 0)) * E
    res.update(step_kernel_launches=len(fk), raw_fetch_bytes_per_launch=raw_f, raw_write_bytes_per_launch=raw_w,
               corrected_fetch_bytes_per_launch=raw_f / cf, corrected_write_bytes_per_launch=raw_w / cw,
               traffic_bytes_per_launch=raw_f / cf + raw_w / cw,
               algorithmic_read_bytes=alg_read, algorithmic_write_bytes=alg_write,
               algorithmic_bytes_per_launch=alg_read + alg_write,
               traffic_over_algorithmic=(raw_f / cf + raw_w / cw) / (alg_read + alg_write))
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
