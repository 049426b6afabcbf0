"""Differential fuzzing, GPU path vs CPU oracle (hypothesis): arbitrary Unicode text incl. astral planes,
combining marks, empty strings, and long runs -- bit-exact for all five measures (SURVEY.md section 4 asks for it)."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

import oracle_lib as O

pytestmark = pytest.mark.gpu

text = st.text(alphabet=st.characters(blacklist_categories=("Cs",)), min_size=0, max_size=60)
ascii_text = st.text(alphabet=st.characters(min_codepoint=1, max_codepoint=127), min_size=0, max_size=140)
pairs = st.lists(st.tuples(st.one_of(text, ascii_text), st.one_of(text, ascii_text)), min_size=1, max_size=120)


@pytest.fixture(scope="module")
def ctx():
    import strsim_amd as S
    c = S.Context(0)
    yield S, c
    c.close()


@settings(max_examples=60, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(rows=pairs, measure=st.sampled_from(O.MEASURES))
def test_gpu_equals_oracle(ctx, rows, measure):
    S, c = ctx
    A = [r[0] for r in rows]
    B = [r[1] for r in rows]
    ao, av = S.pack_strings(A)
    bo, bv = S.pack_strings(B)
    got = c.pairs_host(measure, ao, av, bo, bv)
    exp = O.batch_strings(measure, A, B)
    bad = np.nonzero(got.view(np.uint64) != exp.view(np.uint64))[0]
    assert bad.size == 0, (measure, A[bad[0]], B[bad[0]], got[bad[0]], exp[bad[0]])
