bash tools/variants.sh default prev
