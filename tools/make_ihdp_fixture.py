#!/usr/bin/env python3
"""Writes tests/golden/ihdp_covariates.npz from the IHDP data file the reference's simulation harness holds
(reference ihdp/sim.data.gz), following ihdp/data.R:1-22: drop treated children of non-white mothers (747 rows remain),
6 continuous + 19 binary covariates, treatment z, grouping factors g1 = mother's age clipped to [15, 40] (26 levels) and
g2 = site.  Run in the build container (the reference tree is not on the GPU box); the output is data, not source."""
import gzip
import os

import numpy as np
import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = "/root/reference/ihdp/sim.data.gz"
COVS_CONT = ["bw", "b.head", "preterm", "birth.o", "nnhealth", "momage"]
COVS_CAT = ["sex", "twin", "b.marr", "mom.lths", "mom.hs", "mom.scoll", "cig", "first", "booze", "drugs", "work.dur", "prenatal",
            "ark", "ein", "har", "mia", "pen", "tex", "was"]

with gzip.open(SRC, "rt") as f:
    df = pd.read_csv(f, sep="\t")
df = df[(df["treat"] != 1) | (df["momwhite"] != 0)]
x = df[COVS_CONT + COVS_CAT].to_numpy(dtype=np.float64)
g1 = df["momage"].to_numpy().copy()
g1[g1 < 16] = 15
g1[g1 > 39] = 40
levels = np.unique(g1)
out = os.path.join(ROOT, "tests", "golden", "ihdp_covariates.npz")
np.savez_compressed(out, x=x, z=df["treat"].to_numpy(dtype=np.float64), g1=(np.searchsorted(levels, g1) + 1).astype(np.int32),
                    g2=df["site.num"].to_numpy(dtype=np.int32), names=np.array(COVS_CONT + COVS_CAT))
print(out, x.shape, "g1 levels", len(levels), "g2 levels", len(np.unique(df["site.num"])), "treated", int(df["treat"].sum()))
