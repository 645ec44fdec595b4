cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r06_sites; mkdir -p $O
rm -f gpurun_out/precision.jsonl
for v in s0 s1 s2 s3 s4 s5 s6 s7; do
  DSDF_LIB_PATH=$PWD/differentiable-sdf-rendering_amd/lib/variants/libdsdf_$v.so timeout 600 python -m pytest tests/test_refshim_fixture.py -q -m gpu -p no:cacheprovider -k "within_the_reference_fp32_floor" 2>&1 | tail -1
done
cp gpurun_out/precision.jsonl $O/precision_sites.jsonl
