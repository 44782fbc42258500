# round 6: the END-TO-END accuracy of the shipped path as numbers, from the GPU parity tests themselves — each figure with
# the F(4x4) kernel under its default layer rule and with it off (ADM_WINO6=0: F(2x2,3x3) everywhere) -> profiles/r06_accuracy.md
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/${1:-r06acc}; mkdir -p $O; rm -f $O/figures.jsonl
K="test_unet_256_matches_the_oracle_at_batch_1 or complete_ddim50 or config1_64x64 or last_steps_match or ddim_10_steps or eta_1 or config4_latent"
for w in default 0; do
  if [ $w = default ]; then unset ADM_WINO6; else export ADM_WINO6=$w; fi
  ADM_ACCURACY_LOG=$R/$O/figures.jsonl timeout 1500 python -m pytest tests/test_full_size.py -q -m gpu -k "$K" 2>&1 | tail -3 | tee -a $O/pytest.txt
done
python tools/format_accuracy.py $O/figures.jsonl > $O/r06_accuracy_table.md; cat $O/r06_accuracy_table.md
