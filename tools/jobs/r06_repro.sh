cd $GRAFT_REPO_ROOT
# the round-end sequence on the final tree: the GPU suite, smoke(), the default bench line; and where a key upload's time goes
O=gpurun_out/r6x_last; mkdir -p $O
(timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3) | tee $O/pytest_gpu.txt
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1) | tee $O/smoke.txt
(GS_HOST_TRACE=1 timeout 600 python tools/time_key_upload.py 20 2>&1 | grep -v "amdgpu.ids\|gs host" ) | tee $O/key_upload.txt
(timeout 900 python bench.py 2>/dev/null | tail -1) > $O/bench_line.json; head -c 400 $O/bench_line.json; echo
