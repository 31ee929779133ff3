bash tools/variants.sh --config 2 -k 21 -- default k21w5
