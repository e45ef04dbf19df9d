R=$PWD; mkdir -p gpurun_out
{
echo "== whole-call A/B (us): new, ablations"
timeout 900 python tools/mb_attn_bwd_ab.py new= $(for v in "$@"; do echo -n "a$v=HOISDF_LIB=ab/lib_b4a$v.so "; done)
} > gpurun_out/bwd4_abl2.log 2>&1
cat gpurun_out/bwd4_abl2.log
