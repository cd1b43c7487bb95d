"""Refresh the FM entries of profiles/rNN/traffic_rNN.json from the per-dispatch PMC averages that profiles/pmc.py printed
(pmc_fm_tierc{0,1}_{FETCH,WRITE}_SIZE.txt of the same directory): hbm_bytes_per_launch = (2 x FETCH_SIZE + WRITE_SIZE) KiB
(MI355X_MICROARCH.md, HBM section: gfx950 tallies 128-byte requests at 64 bytes).
    python profiles/traffic.py profiles/r04"""
import json
import os
import re
import sys

KEYS = {  # entry -> (tier-C setting of the pass, kernel substring)
    "fm": (0, "fm_fused_fwd_kernel"), "fm_segment_reduce_tier_b": (0, "segment_reduce_kernel<rbx::FmPolicy"),
    "fm_rezero": (0, "rezero_rows_kernel"), "fm_ta_reduce": (0, "ta_reduce_lds_kernel"), "fm_ta_final": (0, "ta_final_kernel"),
    "fm_compact_ids": (0, "compact_ids_kernel"), "fm_tier_c_reduce": (1, "tc_reduce_kernel"),
    "fm_tier_c_scatter": (1, "tc_scatter_kernel"), "fm_tier_c_count": (1, "tc_count_kernel"),
    "fm_tier_c_rezero": (1, "tc_rezero_kernel"),
}


def read(path):
    out = {}
    for line in open(path):
        m = re.match(r"(?:void )?(.*?)\s+dispatches\s+\d+\s+\S+ per dispatch\s+([0-9.]+)", line)
        if m:
            out[m.group(1).strip()] = float(m.group(2))
    return out


def main():
    d = sys.argv[1]
    tag = os.path.basename(os.path.normpath(d))
    path = os.path.join(d, "traffic_%s.json" % tag)
    doc = json.load(open(path))
    vals = {(tc, c): read(os.path.join(d, "pmc_fm_tierc%d_%s.txt" % (tc, c))) for tc in (0, 1) for c in ("FETCH_SIZE", "WRITE_SIZE")}
    for key, (tc, sub) in KEYS.items():
        f = [v for k, v in vals[(tc, "FETCH_SIZE")].items() if sub in k]
        w = [v for k, v in vals[(tc, "WRITE_SIZE")].items() if sub in k]
        if key in doc and f and w:
            doc[key]["FETCH_SIZE_KB_raw"] = f[0]
            doc[key]["WRITE_SIZE_KB"] = w[0]
            doc[key]["hbm_bytes_per_launch"] = int(round((2 * f[0] + w[0]) * 1024))
    json.dump(doc, open(path, "w"), indent=1)
    for key in KEYS:
        if key in doc:
            print(key, doc[key]["hbm_bytes_per_launch"])


if __name__ == "__main__":
    main()
