"""Writes tests/golden/rng_kat.json.

The reference holds no golden vectors for this path and cannot be imported or
built here, so the only values that can be pinned are the published known answers of the two
third-party generators it draws from:
  * R >= 3.6 (Mersenne-Twister / Inversion / Rejection): values every R installation prints, as listed
    in SURVEY.md Appendix B.1 (7 significant digits), plus `set.seed(1); rexp(3)`.
  * boost::ecuyer1988: Boost's own validation constant (libs/random/test/test_ecuyer1988.cpp):
    the 10000th output of a default-constructed engine is 2060321752.
This script only records those literals; nothing is generated from repository code.
"""
import json
import os

KAT = {
    "runif": {"1": [0.2655087, 0.3721239, 0.5728534], "42": [0.9148060, 0.9370754, 0.2861395],
              "123": [0.2875775, 0.7883051, 0.4089769]},
    "rnorm": {"1": [-0.6264538, 0.1836433, -0.8356286], "42": [1.3709584, -0.5646982, 0.3631284],
              "123": [-0.56047565, -0.23017749, 1.55870831]},
    "rexp": {"1": [0.7551818, 1.1816428, 0.1457067]},
    "sample10": {"42": [1, 5, 10, 8, 2, 4, 6, 9, 7, 3], "123": [3, 10, 2, 8, 6, 9, 1, 7, 5, 4]},
    "ecuyer1988_default_10000th": 2060321752,
}

if __name__ == "__main__":
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "rng_kat.json"), "w") as f:
        json.dump(KAT, f, indent=1)
