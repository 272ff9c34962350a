import contextlib, io, os, sys, warnings
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from gaussian_processes_amd import utils as gp
import test_gpu_dropin as t
g = np.load("tests/golden/g6_vargp_full_N128.npz")
fit, err, R = t._run_vargp(gp, g)
print("err", err["is_error"], err.get("error_message"))
print("mine", fit["values_track"]["loss_track"]["logmarginal"].numpy())
print("ref ", g["logmarginal"])
print("KL mine", fit["values_track"]["loss_track"]["KL"].numpy(), "ref", g["KL"])
print("logA mine", fit["values_track"]["f_par_track"]["logA"].numpy(), "ref", g["logA_track"])
print("l0 mine", fit["values_track"]["f_par_track"]["lambda0"].numpy(), "ref", g["lambda0_track"])
