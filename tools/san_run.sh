# Host-side ASan + UBSan run of the C host (renderer.c, pt_image_io.c, scene_driver.c) and the host half of
# pt_kernels.hip, linked into one executable (build/san/pt_scene_driver_san, built by the hipcc lines in DESIGN.md §6);
# device code is NOT instrumented.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/san && cd gpurun_out/san
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
B=$GRAFT_REPO_ROOT/build/san/pt_scene_driver_san
echo "== plain"; PT_NUM_GPUS=1 PT_PIPELINE=0 timeout -k 5 120 $B 3 2 96 54 8 50 plain%d.png > plain.log 2>&1 || { tail -30 plain.log; exit 1; }
echo "== pipeline"; PT_NUM_GPUS=1 PT_PIPELINE=2 timeout -k 5 120 $B 3 5 96 54 8 50 pipe%d.png > pipe.log 2>&1 || { tail -30 pipe.log; exit 1; }
echo "== three ranks on one GPU"; PT_NUM_GPUS=3 PT_SHARE_GPU=1 PT_PIPELINE=0 timeout -k 5 120 $B 3 2 96 54 8 50 ranks%d.png > ranks.log 2>&1 || { tail -30 ranks.log; exit 1; }
echo "== three ranks + pipeline"; PT_NUM_GPUS=3 PT_SHARE_GPU=1 PT_PIPELINE=2 timeout -k 5 120 $B 3 4 96 54 8 50 rp%d.png > rp.log 2>&1 || { tail -30 rp.log; exit 1; }
echo "== eight ranks + pipeline 4, 640x360"; PT_NUM_GPUS=8 PT_SHARE_GPU=1 PT_PIPELINE=4 timeout -k 5 160 $B 10 6 640 360 4 50 big%d.png > big.log 2>&1 || { tail -30 big.log; exit 1; }
echo "== grid + relaxed math switches"; PT_ACCEL=grid PT_MATH=fast PT_NUM_GPUS=1 timeout -k 5 120 $B 1 1 96 54 4 50 sw%d.bmp > sw.log 2>&1 || { tail -30 sw.log; exit 1; }
echo "== jpg + bmp names"; PT_NUM_GPUS=1 timeout -k 5 120 $B 0 1 64 36 4 50 out%d.jpg > jpg.log 2>&1 || { tail -30 jpg.log; exit 1; }
cmp plain3.png ranks3.png && cmp plain4.png pipe4.png && echo "images agree"
grep -l "ERROR\|runtime error" *.log; echo "sanitizer run done"
