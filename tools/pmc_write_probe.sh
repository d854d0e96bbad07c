#!/bin/bash
# WRITE_SIZE / FETCH_SIZE of the conv kernels for a few layers (filter list): per-dispatch averages
REPO=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp
for f in "l4.0" "l3.0" "nlc_c2" "c1_2"; do
  for c in WRITE_SIZE FETCH_SIZE; do
    rm -rf /tmp/pw_$c
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pw_$c -o p -- python $REPO/tools/bench_layers.py --filter "$f" --iters 1 > /dev/null 2>&1
    python3 - "$(find /tmp/pw_$c -name '*counter_collection.csv' | head -1)" "$f" $c <<'PY'
import csv,sys,re,collections
acc=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    k=r["Kernel_Name"]
    m=re.search(r"(k_\w+<[^>]*>)",k)
    if m and ("conv" in k or "wgrad" in k): acc[m.group(1)].append(float(r["Counter_Value"]))
for k,v in acc.items(): print(sys.argv[2], sys.argv[3], k, "avg KB %.0f"%(sum(v)/len(v)), "n", len(v))
PY
  done
done
