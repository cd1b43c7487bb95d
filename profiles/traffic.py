"""Write / refresh profiles/rNN/traffic_rNN.json from the per-dispatch PMC averages that profiles/pmc.py printed
(pmc_<config>_{FETCH,WRITE}_SIZE.txt of the same directory; separate rocprofv3 --pmc passes):
hbm_bytes_per_launch = (2 x FETCH_SIZE + WRITE_SIZE) KiB (MI355X_MICROARCH.md, HBM section: gfx950 tallies 128-byte
requests at 64 bytes).  An entry whose "kernel" lists several kernels (one C-ABI call that launches two) sums them.
    python profiles/traffic.py profiles/r05"""
import json
import os
import re
import sys

# entry -> (config of the pass, kernel substrings, the name bench.py's roofline uses or a description, algorithmic bytes or None)
KEYS = {
    "fm": ("fm", ["fm_quad_fwd_kernel"], "fm_quad_fwd_kernel<10,20,3>", 133169152),
    "fm_segment_reduce_tier_b": ("fm", ["segment_reduce_kernel<rbx::FmPolicy"],
                                 "segment_reduce_kernel<FmPolicy,4,1,true> (the 12 large tables)", 132000000),
    "fm_rezero": ("fm", ["rezero_rows_kernel"], "rezero_rows_kernel<true>", None),
    "fm_ta_reduce": ("fm", ["ta_reduce_lds_kernel"], "ta_reduce_lds_kernel<4>", None),
    "fm_ta_final": ("fm", ["ta_final_kernel"], "ta_final_kernel<4,true>", None),
    "fm_compact_ids": ("fm", ["compact_ids_kernel"], "compact_ids_kernel<true>", None),
    "fm_radix_scatter": ("fm", ["radix_scatter_kernel"], "radix_scatter_kernel<8> (one of three passes)", None),
    "youtubednn": ("youtubednn", ["embed_seq_kernel", "embed_fwd_kernel"],
                   "embed_seq_kernel<64,2,1,true> + embed_fwd_kernel<32,1,true> (one rbx_embed_fwd call)", 1240834022),
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
    doc = json.load(open(path)) if os.path.exists(path) else {}
    doc["_how"] = ("rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE in SEPARATE passes (profiles/scripts/%s_final.sh: "
                   "bench.py --no-cpu-baseline [--config C] --eager --steps 5 --warmup 3), per-dispatch averages via profiles/pmc.py "
                   "(pmc_<config>_{FETCH,WRITE}_SIZE.txt); FETCH_SIZE x2 (gfx950 tallies 128-B requests at 64 B: MI355X_MICROARCH.md, "
                   "HBM section), WRITE_SIZE as is; unit KiB; 8 distinct batches rotated." % tag)
    cache = {}
    for key, (cfg, subs, name, algo) in KEYS.items():
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            f = os.path.join(d, "pmc_%s_%s.txt" % (cfg, c))
            if (cfg, c) not in cache:
                cache[(cfg, c)] = read(f) if os.path.exists(f) else {}
        fs = [sum(v for k, v in cache[(cfg, "FETCH_SIZE")].items() if s in k) for s in subs]
        ws = [sum(v for k, v in cache[(cfg, "WRITE_SIZE")].items() if s in k) for s in subs]
        if not all(fs) or not all(ws):
            continue
        ent = doc.setdefault(key, {})
        ent["kernel"] = name
        ent["FETCH_SIZE_KB_raw"] = round(sum(fs), 1)
        ent["WRITE_SIZE_KB"] = round(sum(ws), 1)
        ent["hbm_bytes_per_launch"] = int(round((2 * sum(fs) + sum(ws)) * 1024))
        if algo:
            ent["algorithmic_bytes_per_launch"] = algo
    json.dump(doc, open(path, "w"), indent=1)
    for key in KEYS:
        if key in doc:
            print(key, doc[key]["hbm_bytes_per_launch"], doc[key].get("algorithmic_bytes_per_launch"))


if __name__ == "__main__":
    main()
