for v in 0 1 2; do echo "SF_HBM_PLANE_K_MAX=$v"; SF_HBM_PLANE_K_MAX=$v python3 tools/r04_loop_probe.py own 2>&1 | tail -2; done
SF_HBM_PLANE_K_MAX=1 timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lockstep" 2>&1 | tail -2
