"""Summarises rocprofv3 counter_collection.csv files: per kernel name (and per launch ordinal within a frame), mean counter values.
usage: pmc_summary.py DIR [DIR ...]  (each DIR = one --pmc pass)"""
import csv, glob, json, os, sys
from collections import defaultdict, OrderedDict
out = OrderedDict()
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per = defaultdict(lambda: defaultdict(float)); seen = defaultdict(set); order = []
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"].split("(")[0]
            if not (name.startswith("void k_") or name.startswith("k_")): continue
            key = name.replace("void ", "")
            did = int(row["Dispatch_Id"])
            per[(key, did)][row["Counter_Name"]] += float(row["Counter_Value"])
        # group dispatches of the same kernel by their ordinal within the frame
        bykernel = defaultdict(list)
        for (key, did) in sorted(per, key=lambda k: k[1]): bykernel[key].append(did)
        nframes = int(os.environ.get("RTX_PMC_FRAMES", "3"))
        for key, dids in bykernel.items():
            per_frame = max(1, len(dids) // nframes)
            for o in range(per_frame):
                sel = dids[o::per_frame]
                acc = defaultdict(float)
                for did in sel:
                    for c, v in per[(key, did)].items(): acc[c] += v / len(sel)
                out.setdefault(f"{key}#{o}", {}).update({c: round(v, 1) for c, v in acc.items()})
print(json.dumps(out, indent=1))
