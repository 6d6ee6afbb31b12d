cp ab/libblsmi_new.so bls_amd/libblsmi.so; touch bls_amd/libblsmi.so
python -m pytest tests/test_gpu_pairing.py tests/test_gpu_round2.py -x -q 2>&1 | grep -E "passed|failed|rror" | head -5
for v in old new old new; do
  cp ab/libblsmi_$v.so bls_amd/libblsmi.so; touch bls_amd/libblsmi.so
  python bench.py --no-cpu-baseline --no-ref-shapes --no-verify-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['roofline']['kernel_ms'])"
done
