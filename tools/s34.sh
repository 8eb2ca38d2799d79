timeout 600 python -m pytest tests/test_gpu_cells.py -x -q 2>&1 | tail -2
python bench_ops.py --only sacell --out gpurun_out/s34_ops.json 2>&1 | grep "sa_cell "
for t in _l0 _n1; do python tools/sa_cell_probe.py $t 100 2>&1 | grep shape; done
