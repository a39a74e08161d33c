for v in 8 1; do echo "SF_RANK_K_MIN=$v"; SF_RANK_K_MIN=$v python3 tools/r04_loop_probe.py own 2>&1 | tail -2; done
