#!/bin/bash
# A/B of environment knobs over tools/bench_layers.py rows, side by side (fwd / dgrad / wgrad ms), on the GPU box:
#   bash tools/ab_layers.sh "<row filter or ''>" "" "CFUN_WINO_SB=0" "CFUN_WINO_2D=1 CFUN_WINO_SB=1" ...
# Every argument after the filter is one variant: a (possibly empty) list of VAR=VALUE settings.  The box-to-box and
# run-to-run spread of these rows is 2 - 5 %: repeat a variant ("" "") to see it before reading a difference.
REPO=$(cd "$(dirname "$0")/.." && pwd)
FILT=$1; shift
mkdir -p "$REPO/gpurun_out"
i=0
files=()
for v in "$@"; do
  f="$REPO/gpurun_out/ab_$i.log"
  env $v python "$REPO/tools/bench_layers.py" ${FILT:+--filter "$FILT"} 2>/dev/null | grep -v "amdgpu.ids" | cut -c1-78 > "$f"
  files+=("$f"); echo "variant $i: ${v:-<default>}"; i=$((i + 1))
done
python3 - "${files[@]}" <<'PY'
import sys
cols = [open(f).read().splitlines() for f in sys.argv[1:]]
for rows in zip(*cols):
    print(rows[0][:46] + " | ".join(r[46:78] for r in rows))
PY
