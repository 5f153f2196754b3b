"""which hardware queue each kind of kernel of the LAST engine of a traced bench run went down: python tools/micro/queues_of.py <kernel_trace.csv> [last_n]"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tail = rows[-int(sys.argv[2]) if len(sys.argv) > 2 else -1200:]
agg = collections.Counter()
for r in tail:
    n = r["Kernel_Name"].replace("void ", "").replace("hcv::(anonymous namespace)::", "").replace("hcv::", "")
    agg[(re.sub(r"\(.*", "", n)[:34], r["Grid_Size_X"], r["Queue_Id"])] += 1
byq = collections.defaultdict(list)
for (n, g, q), c in sorted(agg.items()):
    if c >= 4:
        byq[q].append(f"{n}/{g} x{c}")
for q in sorted(byq):
    print("queue", q, "|", "; ".join(byq[q]))
