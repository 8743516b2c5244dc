fails=0
for i in $(seq 1 25); do
  timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "test_multi_device_context" > /tmp/o.txt 2>&1 || { fails=$((fails+1)); echo "--- failure in iteration $i"; grep -E "AssertionError: \(|bz2_mi355x|fault" /tmp/o.txt | head -4 | cut -c1-200; }
done
echo "multi-device: $fails failures of 25"
