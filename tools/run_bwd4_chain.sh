mkdir -p gpurun_out
{
timeout 600 python -m pytest tests/test_gpu_emu.py tests/test_gpu_bench_geometry.py -q -k "attention" 2>&1 | tail -3
for g in 2 16; do HOISDF_EMU_ATTN_BWD_CHAIN=$g timeout 600 python -m pytest tests/test_gpu_emu.py -q -k "attention" 2>&1 | tail -1; done
echo "== whole-call A/B (us): chains of 4 (default), 2, 8, 16, partials (4p), round 3"
timeout 900 python tools/mb_attn_bwd_ab.py g4= g2=HOISDF_EMU_ATTN_BWD_CHAIN=2 g8=HOISDF_EMU_ATTN_BWD_CHAIN=8 g16=HOISDF_EMU_ATTN_BWD_CHAIN=16 part=HOISDF_EMU_ATTN_BWD=4p old=HOISDF_EMU_ATTN_BWD=3
} > gpurun_out/bwd4_chain.log 2>&1
cat gpurun_out/bwd4_chain.log
