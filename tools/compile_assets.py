#!/usr/bin/env python3
"""Compile the reference's DATA files into the assets this package ships.

Runs only where the reference checkout is mounted (this container).  Inputs
(reference: SURVEY.md section 2 row 8, all data, no code):
  assets/mjcf/dyros_tocabi/xml/dyros_tocabi.xml     -> assets/tocabi_model.json
  assets/DeepMimic/processed_data_tocabi_walk.txt   -> assets/mocap_walk_f32.npy
      (3600 x 36; the task loads it with np.genfromtxt and casts to float32,
       reference: tasks/dyros_dynamic_walk.py:112-113)
  assets/Data/obs_mean_fixed.txt, obs_variance_fixed.txt -> assets/obs_norm_f32.npz
      (reference: tasks/dyros_dynamic_walk.py:139-142)
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from isaacgymdyros_amd import model as M  # noqa: E402

REF = os.environ.get("DW_REFERENCE", "/root/reference")
A = os.path.join(REF, "python/IsaacGymEnvs/assets")


def main():
    if not os.path.isdir(A):
        sys.exit("reference assets not found at %s" % A)
    os.makedirs(M.ASSET_DIR, exist_ok=True)
    mdl = M.compile_mjcf(os.path.join(A, "mjcf/dyros_tocabi/xml/dyros_tocabi.xml"))
    with open(M.MODEL_JSON, "w") as f:
        json.dump(mdl, f, indent=1)
    mocap = np.genfromtxt(os.path.join(A, "DeepMimic/processed_data_tocabi_walk.txt"), encoding="ascii")
    assert mocap.shape == (3600, 36), mocap.shape
    np.save(os.path.join(M.ASSET_DIR, "mocap_walk_f32.npy"), mocap.astype(np.float32))
    mean = np.genfromtxt(os.path.join(A, "Data/obs_mean_fixed.txt"), encoding="ascii")
    var = np.genfromtxt(os.path.join(A, "Data/obs_variance_fixed.txt"), encoding="ascii")
    assert mean.shape == (37,) and var.shape == (37,)
    np.savez(os.path.join(M.ASSET_DIR, "obs_norm_f32.npz"),
             mean=mean.astype(np.float32), var=var.astype(np.float32))
    print("bodies", len(mdl["body_names"]), "geoms", len(mdl["geoms"]),
          "mass", sum(mdl["inert_mass"]))


if __name__ == "__main__":
    main()
