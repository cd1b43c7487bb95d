cd /root/repo; mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | tail -3 > gpurun_out/final_test.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 >> gpurun_out/final_test.txt
