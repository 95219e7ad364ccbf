mkdir -p gpurun_out/r4/sw
for cf in 0.8 0.9 1.0; do for w2 in 2.2 2.6 3.3; do for pt2 in 0.12 0.2; do
t="HYBRID_CELL_FACTOR=$cf,HYBRID_WORK2=$w2,POOL_PIECE_TIME2=$pt2"
TUNE=$t OWN=modular WORLDS=8 FRAMES=40 python scripts/r4_ranks.py 2>/dev/null | grep '^4K' | sed "s/^/$t /" >> gpurun_out/r4/sw/grid.log
done; done; done
