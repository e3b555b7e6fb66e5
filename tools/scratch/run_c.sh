python -m pytest tests/test_gpu_ops.py -x -q -k "sliced_tiles or bit_reproducible or conv2d_forward" 2>&1 | tail -3
B="python bench.py --data-size 1152 --no-cpu-baseline --no-roofline"
for i in 1 2; do
echo "nofold lib:"; IGAN_LIB=$PWD/inclusivegan_amd/csrc/libigan_hip_nofold.so $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['hip_graphs'])"
echo "fold off:";   IGAN_CONV_FOLD_FIXUP=0 $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['hip_graphs'])"
echo "fold <=8:";   $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['hip_graphs'])"
echo "fold <=4:";   IGAN_CONV_FOLD_MAX=4 $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['hip_graphs'])"
done
