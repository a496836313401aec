# smoke(), the new interleaving test, and two more soaks with other seeds on the final library.
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4y; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.txt
python -m pytest tests/test_gpu_prove.py -m gpu -q -k "interleavings or realistic" 2>&1 | tail -2 | tee $O/pytest_new.txt
timeout 400 python tools/soak_mixed.py 280 2 > $O/soak_mixed_seed2.txt 2>&1; echo "exit $?" >> $O/soak_mixed_seed2.txt; tail -3 $O/soak_mixed_seed2.txt | head -2; grep "^OK\|MISMATCH" $O/soak_mixed_seed2.txt
timeout 200 python tools/stress_msm_random.py 100 11 > $O/stress_msm_seed11.txt 2>&1; echo "exit $?" >> $O/stress_msm_seed11.txt; tail -3 $O/stress_msm_seed11.txt
