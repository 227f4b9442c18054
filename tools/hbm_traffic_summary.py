"""FETCH_SIZE / WRITE_SIZE passes (tools/hbm_traffic.sh) -> bytes per launch of each frame kernel.
rocprofv3 reports both counters in KB.  gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE tallies 64 B per 128-byte request of
16-byte-per-lane loads, so read bytes = 2 x FETCH_SIZE (an upper bound for narrower accesses); WRITE_SIZE is exact.  Infinity-Cache hits
are included in both, so the sum is an upper bound on true HBM traffic."""
import csv, glob, json, os, sys
from collections import defaultdict

NAMES = {"k_packet<true": "k_trace_any", "k_trace_fast<true": "k_trace_any", "k_packet<false": "k_trace_closest", "k_trace_fast<false": "k_trace_closest",
         "k_items": "k_trace_items", "k_shade": "k_shade", "k_resolve": "k_resolve"}


def collect(d, counter):
    acc, n = defaultdict(float), defaultdict(int)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter: continue
            k = row["Kernel_Name"].replace("void ", "")
            for pat, name in NAMES.items():
                if k.startswith(pat):
                    acc[name] += float(row["Counter_Value"]); n[name] += 1
                    break
    return acc, n


fetch, nf = collect(sys.argv[1], "FETCH_SIZE")
write, nw = collect(sys.argv[2], "WRITE_SIZE")
out = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `python3 bench.py --steps 6 --warmup 2 --cpu-seconds 0 --no-extras` "
               "(3 frames in flight, one stream per frame: 4 closest-hit launches, ONE shadow-ray packet launch + its item launch, 4 shade, 4 resolve per frame), averaged per launch; "
               "bytes = 2 x FETCH_SIZE (gfx950 correction) + WRITE_SIZE, both reported in KB; Infinity-Cache hits included (upper bound on HBM traffic)"}
import hashlib
_h = hashlib.sha256()
_csrc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cpu-raytracer_amd", "csrc")
for _f in sorted(glob.glob(os.path.join(_csrc, "*.h")) + glob.glob(os.path.join(_csrc, "*.hip"))):
    _h.update(open(_f, "rb").read())
out["sources_sha16"] = _h.hexdigest()[:16]        # bench.py compares it with the sources it runs (roofline.traffic_measured_on_these_sources)
for k in ("k_trace_any", "k_trace_items", "k_trace_closest", "k_shade", "k_resolve"):
    if nf[k] and nw[k]:
        f, w = fetch[k] / nf[k], write[k] / nw[k]
        out[k + "_bytes_per_launch"] = int((2 * f + w) * 1024)
        out[k + "_fetch_KB_raw"] = round(f, 1); out[k + "_write_KB_raw"] = round(w, 1); out[k + "_launches_sampled"] = nf[k]
print(json.dumps(out, indent=1))
