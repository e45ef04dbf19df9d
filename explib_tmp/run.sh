cd /root/repo
X=explib_tmp
echo "fwd/dX us for (65536,1024,256) (65536,256,1024) (65536,768,256) (65536,256,256) (294912,256,256) (49152,1024,992) (49152,512,512) (49152,256,256)"
for rep in 1 2 3; do
TAG=base HOISDF_LIB=$X/libhoisdf_base.so python $X/exp.py
TAG=new python $X/exp.py
done
python -m pytest tests/test_gpu_emu.py -x -q 2>&1 | tail -3
