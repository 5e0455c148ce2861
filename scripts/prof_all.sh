#!/bin/bash
# All profile passes behind profiles/ (run on the GPU box through gpurun): kernel stats of the bench command (3 x 3 Gbp), HBM
# traffic (FETCH_SIZE and WRITE_SIZE in separate passes), SQ instruction counters (three passes), kernel stats of config 4 on
# one GPU and of the end-to-end run.  Each rocprofv3 run uses --kernel-trace only next to --pmc.  Summaries land in
# gpurun_out/prof/ and are copied into profiles/ by hand.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof
mkdir -p $O
LEGS="--no-cpu-baseline --no-e2e --no-c4-leg --no-cold-leg --no-nruns-leg --no-c5-leg --no-valley-leg"
B1="python bench.py --steps 1 --warmup 0 $LEGS --no-dense-leg"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python bench.py --steps 5 --warmup 2 $LEGS > $O/stats.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o f -- python bench.py --steps 1 --warmup 0 $LEGS > $O/pf.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o w -- python bench.py --steps 1 --warmup 0 $LEGS > $O/pw.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --kernel-trace --output-format csv -d $O/pmc_sq1 -o s -- $B1 > $O/sq1.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/pmc_sq2 -o s -- $B1 > $O/sq2.log 2>&1
timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $O/pmc_sq3 -o s -- $B1 > $O/sq3.log 2>&1
NTS_PROF_KMERS=3000000000 python profiles/sq_summarize.py $O/pmc_sq1 $O/pmc_sq2 $O/pmc_sq3 > $O/sq_counters.json
F=$(find $O/pmc_fetch -name "*counter_collection.csv" | head -1)
W=$(find $O/pmc_write -name "*counter_collection.csv" | head -1)
python profiles/pmc_summarize.py "$F" "$W" $O/pmc_traffic "python bench.py --steps 1 --warmup 0 $LEGS (3 x 3 Gbp, one launch sequence per genome; pruned step + dense leg + Bloom build)"
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_c3.csv
grep '^{"metric"' $O/stats.log > $O/bench_under_rocprof.json   # the bench line of the profiled run itself: its HIP-event launch time next to rocprofv3's
python profiles/summarize.py $O/kernel_stats_c3.csv "rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 $LEGS (3 x 3 Gbp)" > $O/kernel_stats_c3.md
# config 4 on one GPU (8 x 3 Gbp at 10 %: accepted-list path), kernel stats + SQ counters and traffic of its select kernel
C4="python bench.py --workload c4 --steps 1 --warmup 0 $LEGS --no-dense-leg"
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --kernel-trace --output-format csv -d $O/c4_sq1 -o s -- $C4 > $O/c4sq1.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/c4_sq2 -o s -- $C4 > $O/c4sq2.log 2>&1
timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $O/c4_sq3 -o s -- $C4 > $O/c4sq3.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/c4_f -o f -- $C4 > $O/c4f.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/c4_w -o w -- $C4 > $O/c4w.log 2>&1
NTS_PROF_KMERS=3000000000 NTS_PROF_WORKLOAD="config 4 on one GPU: 8 x 3 Gbp at 10 %, one genome per launch" python profiles/sq_summarize.py $O/c4_sq1 $O/c4_sq2 $O/c4_sq3 > $O/c4_sq_counters.json
python profiles/pmc_summarize.py "$(find $O/c4_f -name '*counter_collection.csv' | head -1)" "$(find $O/c4_w -name '*counter_collection.csv' | head -1)" $O/c4_pmc_traffic "$C4 (8 x 3 Gbp at 10 % on one GPU)"
rm -rf $O/c4_sq1 $O/c4_sq2 $O/c4_sq3 $O/c4_f $O/c4_w
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats4 -o s -- python bench.py --workload c4 --steps 2 --warmup 1 $LEGS --no-dense-leg > $O/stats4.log 2>&1
find $O/stats4 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_c4.csv
python profiles/summarize.py $O/kernel_stats_c4.csv "rocprofv3 --kernel-trace --stats -- python bench.py --workload c4 --steps 2 --warmup 1 (8 x 3 Gbp at 10 % on one GPU)" > $O/kernel_stats_c4.md
# the assembly-like family (the c5_like leg's): kernel stats of sketch + Bloom build
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats5 -o s -- python bench.py --family assembly-like --divergence 0.013 --steps 3 --warmup 1 $LEGS --no-dense-leg > $O/stats5.log 2>&1
find $O/stats5 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_c5_like.csv
python profiles/summarize.py $O/kernel_stats_c5_like.csv "rocprofv3 --kernel-trace --stats -- python bench.py --family assembly-like --divergence 0.013 --steps 3 --warmup 1 (3 x 3 Gbp assembly-like genomes)" > $O/kernel_stats_c5_like.md
rm -rf $O/stats5
# config 4's cascade level by level (the build against the literal level over a sparse running filter) and the kernels of a run with a
# ninth genome, whose level the library takes the literal way
timeout 300 python scripts/c4_levels.py --force-from 4 > $O/c4_levels.json 2> $O/c4_levels.log
timeout 300 python scripts/c4_levels.py --genomes 9 --force-from 8 --modes "build,auto,forced" > $O/c4_levels_9genomes.json 2>> $O/c4_levels.log
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_lv -o s -- python scripts/c4_levels.py --genomes 9 --modes auto > $O/stats_lv.log 2>&1
find $O/stats_lv -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_c4_levels.csv
python profiles/summarize.py $O/kernel_stats_c4_levels.csv "rocprofv3 --kernel-trace --stats -- python scripts/c4_levels.py --genomes 9 --modes auto (8 cascade levels, the last the literal way)" > $O/kernel_stats_c4_levels.md
rm -rf $O/stats_lv
# short windows (w = 10 / 33 / 63 on one 3 Gbp genome: the product's choice, the window tiles, the key array)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_sw -o s -- python scripts/short_windows_bench.py > $O/short_windows.json 2> $O/short_windows.log
find $O/stats_sw -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_short_windows.csv
python profiles/summarize.py $O/kernel_stats_short_windows.csv "rocprofv3 --kernel-trace --stats -- python scripts/short_windows_bench.py (one 3 Gbp genome, w = 10 / 33 / 63, with and without the filter, three ways)" > $O/kernel_stats_short_windows.md
rm -rf $O/stats_sw
# end to end (FASTA files -> TSV)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_e2e -o s -- python scripts/e2e_synth.py > $O/e2e.json 2> $O/e2e.log
find $O/stats_e2e -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_e2e.csv
python profiles/summarize.py $O/kernel_stats_e2e.csv "rocprofv3 --kernel-trace --stats -- python scripts/e2e_synth.py (3 x 3 Gbp FASTA files -> synteny TSV)" > $O/kernel_stats_e2e.md
rm -rf $O/pmc_sq1 $O/pmc_sq2 $O/pmc_sq3 $O/stats $O/stats4 $O/stats_e2e $O/pmc_fetch $O/pmc_write
ls -la $O | tail -20
