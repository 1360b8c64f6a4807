"""Host-side profile of one fit (development probe): where does the non-GPU time go?"""
import sys, time, cProfile, pstats, numpy as np
sys.path.insert(0, '.')
import bigkrls_amd as bk
from bigkrls_amd.synth import synth
ctx = bk.Context(0)
X, y = synth(20000, 20, 103)
out = bk.bigKRLS(y, X, ctx=ctx); del out
pr = cProfile.Profile(); pr.enable()
out = bk.bigKRLS(y, X, ctx=ctx)
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(18)
