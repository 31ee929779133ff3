python -m pytest tests -x -q -m gpu -k "length_range or fastx" 2>&1 | tail -3
