mkdir -p gpurun_out/r5c18
( timeout 600 python -m pytest tests/test_modules_gpu.py -q -k "lagging" 2>&1 | tail -4 ) > gpurun_out/r5c18/with_fix.log 2>&1
echo "--- with the fix:"; tail -2 gpurun_out/r5c18/with_fix.log
# the same test against the bug it is meant to catch: drop the record_stream of the index lists
cp cfun_amd/weights.py /tmp/weights_backup.py
python - <<'PY'
p = "cfun_amd/weights.py"
s = open(p).read()
assert "idx.record_stream(ws.side)" in s
open(p, "w").write(s.replace("idx.record_stream(ws.side)", "pass"))
PY
( timeout 600 python -m pytest tests/test_modules_gpu.py -q -k "lagging" 2>&1 | grep -E "passed|failed|Error|error|exception" | tail -4 ) > gpurun_out/r5c18/without_fix.log 2>&1
cp /tmp/weights_backup.py cfun_amd/weights.py
echo "--- without the record_stream of the index lists:"; tail -4 gpurun_out/r5c18/without_fix.log
