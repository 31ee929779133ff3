python -m pytest tests -x -q -m gpu -k "short_ragged" 2>&1 | tail -3
