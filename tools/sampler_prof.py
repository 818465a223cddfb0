"""cProfile of the device_chunks sampler mode (GPU box): where the host side of a queue cycle goes."""
import cProfile, pstats, sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import sampler_bench
sampler_bench.run("C2", maxcall=30000, modes=("device_chunks",))       # warm-up
cProfile.run('sampler_bench.run("C2", maxcall=300000, modes=("device_chunks",))', '/tmp/sb.prof')
st = pstats.Stats('/tmp/sb.prof')
st.sort_stats('tottime').print_stats(28)
st.sort_stats('cumulative').print_stats('nested|device', 20)
