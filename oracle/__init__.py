"""CPU oracle for the ThePayne nested-sampling likelihood hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``thepayne_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` do, and there only as the checker / the timed CPU baseline.

What it is: a plain numpy (fp64) restatement of the reference's algorithm for
the path ``lnprobfn -> likelihood.lnlikefn -> lnlike -> GenMod.genspec ->
PayneSpecPredict.getspec -> {Net.eval, smoothspec('vsini'), Doppler,
smoothspec('R'), np.interp} -> chi^2 [+ FastPayneSEDPredict.sed]``; every
function cites the reference file:line it follows.

Parity pinning: the reference (pacargile/ThePayne) ships no tests, golden
vectors or fixtures for this path, so the oracle is pinned
against outputs of the reference itself: ``oracle/gen_golden.py`` imports the
unmodified reference from /root/reference (with dynesty/astropy/h5py stubbed at
import time only) and freezes its outputs as ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` checks this restatement against them to
<= 1e-12 (fp64).  The third-party arithmetic at the boundary (numpy fft /
interp, scipy.special.j1) is unpinned by the reference (setup.py pins no
versions); the fixtures record the versions they were generated with.
"""
from .payne_oracle import *  # noqa: F401,F403
