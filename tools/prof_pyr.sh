cd /tmp && export TMPDIR=/tmp
export FPR_OPTS=mg_pyr_down=1
rm -rf /tmp/pp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -- python3 $GRAFT_REPO_ROOT/tools/prof_mg.py 2049 5 jacobi 5 > /tmp/pp_run.txt 2>/tmp/pp.err
python3 $GRAFT_REPO_ROOT/tools/prof_summarize.py tailstats /tmp/pp /tmp/pp_stats.txt k_cycle_init 5
cat /tmp/pp_stats.txt | cut -c1-130
python3 $GRAFT_REPO_ROOT/tools/prof_summarize.py timeline /tmp/pp /tmp/pp_tl.txt 24
cat /tmp/pp_tl.txt | cut -c1-110
