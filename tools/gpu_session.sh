python -m pytest tests/test_cpp_host_layer.py -x -q -m gpu 2>&1 | tail -3
