SIDE0="--hard-steps 0 --spread-steps 0 --other-configs 0 --extras 0 --yfcc-n 0 --cfg5-images 0 --exhaustive-steps 0 --dry-run-shards 0 --no-cpu --gt 0 --big-batch 0 --nbatches 1 --steps 20 --warmup 3"
for b in 2048 4096 6144 8192; do for o in 1 0; do
  r=$(python bench.py --batch $b $SIDE0 --opt passa_q=$o 2>&1 | grep "summary: headline" | head -1 | cut -c17-90)
  echo "batch $b passa_q=$o: $r"
done; done
