# scratch: the command list of the current gpurun call (tools/README.md); the round's profile set is tools/profile_round.sh
O=gpurun_out/r4r; mkdir -p $O
bash tools/variants.sh default rot rotl0 rotl3 rotl7 > $O/variants.txt 2>&1; cat $O/variants.txt
