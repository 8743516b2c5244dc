for fl in small big stream dstream deflate; do
  timeout 400 python tools/fuzz_parity.py 100 31 $fl 2>&1 | grep -v "amdgpu.ids\|RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -2
done
