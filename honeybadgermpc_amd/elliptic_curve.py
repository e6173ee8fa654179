"""Only the constant this path needs from the reference's elliptic_curve.py (:4-5)."""


class Subgroup:
    # order of the BLS12-381 G1/G2 subgroups = the scalar field all shares live in
    BLS12_381 = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
