S=profiles/scripts/r02_merged_step.py
L=gpurun_out/r02b_variants.log
: > $L
python $S 1.0 1000000 cfg2 base >> $L 2>&1
HGX_LIFT_MINWAVES=8 python $S 1.0 1000000 cfg2 minw8 >> $L 2>&1
HGX_MERGED_BUCKET_RECS=2 python $S 1.0 1000000 cfg2 bucket2 >> $L 2>&1
HGX_MERGED_BUCKET_RECS=4 python $S 1.0 1000000 cfg2 bucket4 >> $L 2>&1
HGX_LIFT_BLOCKS=4 python $S 1.0 1000000 cfg2 blocks4 >> $L 2>&1
HGX_MERGED_WINDOW=2048 python $S 1.0 1000000 cfg2 win2k >> $L 2>&1
python $S 1.0 1250000 cfg4 cfg4 >> $L 2>&1
grep -v amdgpu.ids $L
