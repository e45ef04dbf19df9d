# ablation timing of emu_attn_bwd4_kernel (tools/build_variant.sh b4a<bits> attention_emu_bwd4.hip -DBWD4_ABL=<bits>) + kernel-only times
R=$PWD; mkdir -p gpurun_out
{
timeout 600 python -m pytest tests/test_gpu_emu.py tests/test_gpu_bench_geometry.py -q -k "attention" 2>&1 | tail -5
echo "== kernel trace, new form"; bash tools/prof_attn_bwd.sh
cd $R
echo "== whole-call A/B (us): new, old, ablations"
timeout 900 python tools/mb_attn_bwd_ab.py new= old=HOISDF_EMU_ATTN_BWD=3 $(for v in "$@"; do echo -n "a$v=HOISDF_LIB=ab/lib_b4a$v.so "; done)
} > gpurun_out/bwd4_abl.log 2>&1
cat gpurun_out/bwd4_abl.log
