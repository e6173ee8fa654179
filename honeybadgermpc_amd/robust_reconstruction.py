"""Robust open of ONE shared value from futures of the parties' shares
(replaces honeybadgermpc/robust_reconstruction.py:14-30): a decoder round with a batch of one chunk.

Returns (polynomial, set of senders found in error), or (None, None) when all futures resolved without a decision."""
from .batch_reconstruction import fetch_one
from .polynomial import polynomials_over
from .reed_solomon import Algorithm, DecoderFactory, EncoderFactory, IncrementalDecoder, RobustDecoderFactory


def _codecs(point, t):
    family = Algorithm.FFT if point.use_omega_powers else Algorithm.VANDERMONDE
    return (EncoderFactory.get(point, family), DecoderFactory.get(point, family),
            RobustDecoderFactory.get(t, point, algorithm=Algorithm.GAO))


async def robust_reconstruct(field_futures, field, n, t, point, degree):
    state = IncrementalDecoder(*_codecs(point, t), degree, 1, t)
    async for sender, share in fetch_one(field_futures):
        state.add(sender, [share.value])
        if not state.done():
            continue
        (coefficients,), liars = state.get_results()
        return polynomials_over(field)(coefficients), liars
    return None, None
