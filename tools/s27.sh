mkdir -p gpurun_out/s27
timeout 600 python -m pytest tests/test_gpu_cells.py -x -q 2>&1 | tail -2
python bench_ops.py --only sacell --out gpurun_out/s27/ops.json 2>&1 | grep "sa_cell "
