#!/usr/bin/env python
"""profiles/rNN_traffic.json from the FETCH_SIZE / WRITE_SIZE summaries (tools/pmc_summary.py CSVs).
bytes per launch = 2*FETCH_SIZE (gfx950 wide-read correction, MI355X_MICROARCH.md) + WRITE_SIZE, KiB -> B;
a class with several kernels is the sum of their per-launch averages."""
import csv
import json
import re
import sys

# kernel-name patterns per profiler class (regular expressions matched against the start of the demangled name with a leading "void "
# removed; template arguments change from round to round, so they are matched loosely - a class that matches NO kernel of the trace is
# reported on stderr instead of silently becoming 0)
CLASSES = {
    "wgrad": [r"k_wgrad$", r"k_wgrad_ring8$"],
    "pde_prefilter": [r"k_rk2_fwd<false, false>", r"k_rk2_split<", r"k_rk2_pre16", r"k_rk2_x6<", r"k_rk2_x6w"],
    "rk2_fwd": [r"k_rk2_fwd<true, true>", r"k_rk2_split_uni<", r"k_rk2_x6_uni<"],
    "rk2_bwd": [r"k_rk2_bwd", r"k_rk2_split_bwd<", r"k_rk2_fuse_bwd"],
    "pde_bwd": [r"k_pde_jet_bwd", r"k_pde_tangent_bwd", r"k_pde_value_bwd", r"k_pde_fuse_bwd"],
    "pde_fwd": [r"k_pde_jet6_fwd", r"k_pde_jet_fwd", r"k_pde_value_fwd", r"k_pde_tangent_fwd"],
    "app_fwd": [r"k_app_fwd<true>", r"k_app_feat$"],
    "app_bwd": [r"k_app_bwd"],
    "density_fwd": [r"k_density_q"],
    "density_bwd": [r"k_og<24, true>"],
    "density_scatter": [r"k_tile_scatter(_mfma)?<24>", r"k_tile_hist", r"k_tile_scan", r"k_tile_fill"],
    "app_scatter": [r"k_og<48, true>", r"k_tile_scatter(_mfma)?<48>", r"k_tile_hist", r"k_tile_scan", r"k_tile_fill"],
}


def per_launch(path, counter):
    out = {}
    for r in csv.DictReader(open(path)):
        if r["Counter"] == counter:
            out[r["Kernel"]] = float(r["PerLaunch"])
    return out


def main(fetch_csv, write_csv, out_json, note):
    f, w = per_launch(fetch_csv, "FETCH_SIZE"), per_launch(write_csv, "WRITE_SIZE")
    res = {}
    for cls, ks in CLASSES.items():
        tot, hit = 0.0, 0
        for name in sorted(set(f) | set(w)):
            bare = name[5:] if name.startswith("void ") else name
            if any(re.match(k, bare) for k in ks):
                tot += (2.0 * f.get(name, 0.0) + w.get(name, 0.0)) * 1024.0
                hit += 1
        if not hit:
            print(f"make_traffic: class {cls} matches no kernel of the trace ({ks})", file=sys.stderr)
        res[cls] = tot if hit else None
    json.dump({"source": f"{fetch_csv} + {write_csv} ({note})",
               "note": "bytes per launch = 2*FETCH_SIZE (gfx950 wide-read correction) + WRITE_SIZE, KiB->B; classes with several kernels: sum of their per-launch averages",
               "bytes_per_launch": res}, open(out_json, "w"), indent=1)
    print(json.dumps(res, indent=1))


def whole_step(fetch_csv, write_csv, out_json, note, steps, marker):
    """another workload's counter passes (cfg2 / chessboard / segm): HBM bytes of the WHOLE step = sum over every kernel of the trace of
    (2 x FETCH_SIZE + WRITE_SIZE) / steps, and the same per kernel.  `steps` = launches of `marker`, a kernel that runs once per step."""
    tot = {}
    launches = {}
    for path, counter, mul in ((fetch_csv, "FETCH_SIZE", 2.0), (write_csv, "WRITE_SIZE", 1.0)):
        for r in csv.DictReader(open(path)):
            if r["Counter"] == counter:
                tot[r["Kernel"]] = tot.get(r["Kernel"], 0.0) + mul * float(r["Total"]) * 1024.0
                launches[r["Kernel"]] = int(r["Launches"])
    if steps <= 0:
        ms = [v for k, v in launches.items() if re.match(marker, k[5:] if k.startswith("void ") else k)]
        steps = ms[0] if ms else 1
    per_kernel = {k: v / steps for k, v in sorted(tot.items(), key=lambda kv: -kv[1])}
    json.dump({"source": f"{fetch_csv} + {write_csv} ({note})", "steps": steps,
               "note": "HBM bytes per STEP = sum over all kernels of (2*FETCH_SIZE (gfx950 wide-read correction) + WRITE_SIZE) / steps; KiB->B",
               "bytes_per_step": sum(per_kernel.values()), "bytes_per_step_per_kernel": per_kernel}, open(out_json, "w"), indent=1)
    print(json.dumps({"bytes_per_step": sum(per_kernel.values()), "steps": steps}))


if __name__ == "__main__":
    if "--whole-step" in sys.argv:
        a = [x for x in sys.argv[1:] if x != "--whole-step"]
        whole_step(a[0], a[1], a[2], a[3] if len(a) > 3 else "", int(a[4]) if len(a) > 4 else 0, a[5] if len(a) > 5 else r"k_adam|k_prologue")
    else:
        main(sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else "")
