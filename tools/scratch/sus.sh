cd $GRAFT_REPO_ROOT
echo "== short loops"; IMGS=1024 NO_WGRAD=1 python tools/conv_microbench.py b3b 2>&1 | grep fwd; FB_C1G=1 IMGS=1024 NO_WGRAD=1 python tools/conv_microbench.py b3b 2>&1 | grep fwd
echo "== sustained 1 s: igemm / gemm"; SUSTAIN=1 IMGS=1024 NO_WGRAD=1 python tools/conv_microbench.py b3b 2>&1 | grep fwd; SUSTAIN=1 FB_C1G=1 IMGS=1024 NO_WGRAD=1 python tools/conv_microbench.py b3b 2>&1 | grep fwd
echo "== sustained: pipe vs stream (256->1024 fwd, dgrad)"; SUSTAIN=1 IMGS=1024 NO_WGRAD=1 python tools/conv_microbench.py b3a 2>&1 | grep "fwd"; SUSTAIN=1 FB_C1S_PIPE=0 IMGS=1024 NO_WGRAD=1 python tools/conv_microbench.py b3a 2>&1 | grep "fwd"
echo "== short: pipe vs stream"; IMGS=1024 NO_WGRAD=1 python tools/conv_microbench.py b3a 2>&1 | grep "fwd"; FB_C1S_PIPE=0 IMGS=1024 NO_WGRAD=1 python tools/conv_microbench.py b3a 2>&1 | grep "fwd"
echo "== sustained halo4 default / wide: l2g l3g"; SUSTAIN=1 NO_WGRAD=1 python tools/conv_microbench.py l2g l3g 2>&1 | grep "fwd\|dgrad"; SUSTAIN=1 FB_H4_WIDE=8,16 NO_WGRAD=1 python tools/conv_microbench.py l2g l3g 2>&1 | grep "fwd\|dgrad"
