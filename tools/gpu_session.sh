python -m pytest tests -x -q -m gpu -k "blanked or hist" 2>&1 | tail -8
