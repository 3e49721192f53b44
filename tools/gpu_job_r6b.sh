cd $GRAFT_REPO_ROOT
O=gpurun_out/r6b; mkdir -p $O
python tools/rows_stamps.py 19 > $O/stamps_v19_r16.txt 2>&1
python tools/rows_stamps.py 19 0 > $O/stamps_v19_r8.txt 2>&1
CURIOUS_LIB=abtest/r16w2.so python tools/rows_stamps.py 19 > $O/stamps_v19_r16w2.txt 2>&1
python tools/rows_stamps.py 8 > $O/stamps_v8_r16.txt 2>&1
tail -n 12 $O/stamps_v19_r16.txt $O/stamps_v19_r8.txt $O/stamps_v19_r16w2.txt $O/stamps_v8_r16.txt
