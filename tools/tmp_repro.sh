#!/bin/bash
O=gpurun_out/$1; mkdir -p $O
for i in 1 2 3 4 5 6 7 8 9 10 11; do
( ADM_GRAPH_DRAIN=0 ADM_SEGV_BACKTRACE=1 python -X faulthandler=0 -m pytest tests/test_full_size.py tests/test_pipeline.py -m gpu -x -q -p no:faulthandler ) > $O/sub$i.txt 2>&1
echo "run $i: $(grep -c 'fatal signal' $O/sub$i.txt) $(grep 'passed\|failed' $O/sub$i.txt | tail -1)"
if grep -q "fatal signal" $O/sub$i.txt; then grep -A45 "fatal signal" $O/sub$i.txt | cut -c1-220; break; fi
done
