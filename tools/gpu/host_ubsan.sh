# Host code of libfpv_hip.so (launch logic, graph cache, rotation arithmetic, communicator glue) under UBSan on the GPU box, driven by the GPU
# test suite.  Device code is compiled as always (GPU sanitizers are not available on this pool): every `-fsanitize=` sits behind
# `-Xarch_host`.  AddressSanitizer cannot ride along in a process that initialises the GPU: ROCm's ASan runtime intercepts
# hsa_amd_memory_pool_allocate for its device mode and aborts the first pool allocation ("out of memory ... 0x400000 bytes"); the
# host code that needs no device runs under ASan + UBSan in tests/test_sanitizers.py instead.  The sanitized library replaces
# fpyv_amd/libfpv_hip.so in the box's scratch copy only.
#     gpurun --timeout 1200 -- 'bash tools/gpu/host_ubsan.sh'
set -o pipefail
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/ubsan
cp fpyv_amd/libfpv_hip.so gpurun_out/ubsan/libfpv_hip.plain.so
/opt/rocm/bin/hipcc -O1 -g --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -std=c++17 -shared -fPIC -mllvm -amdgpu-kernarg-preload-count=6 \
    -Xarch_host -fsanitize=undefined -Xarch_host -fno-sanitize-recover=undefined -Xarch_host -fno-omit-frame-pointer -Xarch_host -shared-libsan \
    -o fpyv_amd/libfpv_hip.so fpyv_amd/csrc/fpv_hip.hip > gpurun_out/ubsan/build.log 2>&1 || { echo "sanitized build failed"; tail -5 gpurun_out/ubsan/build.log; exit 1; }
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.ubsan_standalone-x86_64.so)
echo "ubsan runtime: $RT"
LD_PRELOAD=$RT UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1:log_path=gpurun_out/ubsan/ubsan \
    timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fp16.py tests/test_gpu_kstep.py tests/test_gpu_boundary.py tests/test_gpu_traversal.py -m gpu -q -x -k "not bench and not plain_c_host and not two_host_threads" > gpurun_out/ubsan/tests.log 2>&1
rc=$?
cp gpurun_out/ubsan/libfpv_hip.plain.so fpyv_amd/libfpv_hip.so
tail -5 gpurun_out/ubsan/tests.log
ls gpurun_out/ubsan/ | grep -E "^ubsan\." | head
echo "host sanitizers over the GPU suite: rc=$rc"
exit $rc
