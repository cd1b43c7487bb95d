cd /root/repo; mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu -k "graph or replay or capture" 2>&1 | tail -3 > gpurun_out/graph_test.txt
python bench.py --config deepfm --steps 20 --warmup 5 --no-cpu-baseline 2> gpurun_out/deepfm.err | cut -c1-200 >> gpurun_out/graph_test.txt
grep -c "AccumulateGrad" gpurun_out/deepfm.err >> gpurun_out/graph_test.txt
python bench.py --no-cpu-baseline 2> gpurun_out/fm.err | cut -c1-200 >> gpurun_out/graph_test.txt
grep -c "AccumulateGrad" gpurun_out/fm.err >> gpurun_out/graph_test.txt
