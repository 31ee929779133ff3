# scratch: the command list of the current gpurun call (tools/README.md); the round's profile set is tools/profile_round.sh
O=$GRAFT_REPO_ROOT/gpurun_out/r7f; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU2 SQ_BUSY_CYCLES --output-format csv -d $O/pmc -o t -- python3 $GRAFT_REPO_ROOT/tools/bench_ragged.py 20000000 31 > $O/bench.txt 2> $O/err.txt
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/r7f/pmc/*counter_collection.csv')[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if 'scan_bitsliced' in r['Kernel_Name'] and int(r['Grid_Size']) > 100000:
        acc[r['Kernel_Name'][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    print(k, {c: round(sum(v) / len(v) / 1e6, 2) for c, v in d.items()}, 'launches', len(next(iter(d.values()))))
PY
tail -4 $O/bench.txt
