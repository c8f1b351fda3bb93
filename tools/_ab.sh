cd /root/repo
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tile_order or tile_mix" 2>&1 | tail -5
python tools/conv_trace.py --level 3 --dump gpurun_out/trace_l3.npz 2>&1 | tail -40
python tools/conv_trace.py --level 2 --dump gpurun_out/trace_l2.npz 2>&1 | tail -40
