import cProfile, pstats, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/tools")
sys.argv = ["fit_stars.py", "--stars", "3"]
import fit_stars
cProfile.run("fit_stars.main()", "/tmp/fs.prof")
st = pstats.Stats("/tmp/fs.prof")
st.sort_stats("cumulative").print_stats(45)
