"""Single-share robust open (reference: honeybadgermpc/robust_reconstruction.py:14-30):
the same codecs with batch_size = 1."""
from .batch_reconstruction import fetch_one
from .polynomial import polynomials_over
from .reed_solomon import Algorithm, DecoderFactory, EncoderFactory, IncrementalDecoder, RobustDecoderFactory


async def robust_reconstruct(field_futures, field, n, t, point, degree):
    codec = Algorithm.FFT if point.use_omega_powers else Algorithm.VANDERMONDE
    enc = EncoderFactory.get(point, codec)
    dec = DecoderFactory.get(point, codec)
    robust_dec = RobustDecoderFactory.get(t, point, algorithm=Algorithm.GAO)
    inc = IncrementalDecoder(enc, dec, robust_dec, degree, 1, t)
    async for idx, d in fetch_one(field_futures):
        inc.add(idx, [d.value])
        if inc.done():
            polys, errors = inc.get_results()
            return polynomials_over(field)(polys[0]), errors
    return None, None
