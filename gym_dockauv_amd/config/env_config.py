"""
Default configuration dictionaries.  The KEY SCHEMA and default values are those of the reference's
``gym_dockauv/config/env_config.py:9-111`` ("config/ stays"), because they parameterise the kernels and because user
configs written for the reference must keep working.  Keys the reference never reads on the step path
(``radius``, ``w_t``, the three ``*_goal_reached_tol`` besides distance, ``radar.freq``, ``config_name``) are
accepted and ignored here too.
"""
import copy
import os

import numpy as np

# env id -> "module:Class"; unlike the reference's table (which points at an empty envs/__init__.py) these resolve
REGISTRATION_DICT = {
    "SimpleDocking3d-v0": "gym_dockauv_amd.envs:SimpleDocking3d",
    "SimpleCurrentDocking3d-v0": "gym_dockauv_amd.envs:SimpleCurrentDocking3d",
    "CapsuleDocking3d-v0": "gym_dockauv_amd.envs:CapsuleDocking3d",
    "CapsuleCurrentDocking3d-v0": "gym_dockauv_amd.envs:CapsuleCurrentDocking3d",
    "ObstaclesDocking3d-v0": "gym_dockauv_amd.envs:ObstaclesDocking3d",
    "ObstaclesCurrentDocking3d-v0": "gym_dockauv_amd.envs:ObstaclesCurrentDocking3d",
    "ObstaclesNoCapDocking3d-v0": "gym_dockauv_amd.envs:ObstaclesNoCapDocking3d",
}

BASE_CONFIG = {
    # general
    "config_name": "DEFAULT_BASE_CONFIG",
    "title": "DEFAULT",
    "log_level": 20,
    "verbose": 1,
    # episode
    "max_timesteps": 1000,
    # simulation
    "t_step_size": 0.10,
    "interval_datastorage": 100,
    "interval_episode_log": 50,
    "save_path_folder": os.path.join(os.getcwd(), "logs"),
    # goal and done
    "max_dist_from_goal": 20,
    "max_attitude": 60 / 180 * np.pi,
    "dist_goal_reached_tol": 0.5,
    "velocity_goal_reached_tol": 0.3,
    "ang_rate_goal_reached_tol": 20 * np.pi / 180,
    "attitude_goal_reached_tol": 20 * np.pi / 180,
    # vehicle and rewards
    "vehicle": "BlueROV2",
    "u_max": 2.0,
    "v_max": 1.5,
    "w_max": 1.5,
    "p_max": 90 * np.pi / 180,
    "q_max": 90 * np.pi / 180,
    "r_max": 120 * np.pi / 180,
    "radius": 0.5,
    "reward_set": 1,
    "reward_factors": {
        "w_d": 1.1,
        "w_delta_psi": 0.5,
        "w_delta_theta": 0.3,
        "w_phi": 0.3,
        "w_theta": 0.3,
        "w_Thetadot": 0.2,
        "w_t": 0.05,
        "w_oa": 0.20,
        "w_goal": 400.0,
        "w_deltad_max": -200.0,
        "w_Theta_max": -200.0,
        "w_t_max": -100.0,
        "w_col": -300.0,
    },
    "action_reward_factors": 6.0,
    # ray fan, splatted into the radar constructor
    "radar": {
        "freq": 1,
        "alpha": 60 * np.pi / 180,
        "beta": 80 * np.pi / 180,
        "ray_per_deg": 10 * np.pi / 180,
        "max_dist": 10,
        "blocksize_reduce": 2,
    },
}

TRAIN_CONFIG = copy.deepcopy(BASE_CONFIG)
TRAIN_CONFIG["title"] = "Training Run"
TRAIN_CONFIG["save_path_folder"] = os.path.join(os.getcwd(), "logs")

PREDICT_CONFIG = copy.deepcopy(BASE_CONFIG)
PREDICT_CONFIG["interval_datastorage"] = 1
PREDICT_CONFIG["title"] = "Prediction Run"
PREDICT_CONFIG["save_path_folder"] = os.path.join(os.getcwd(), "predict_logs")
PREDICT_CONFIG["interval_episode_log"] = 1

MANUAL_CONFIG = copy.deepcopy(BASE_CONFIG)
MANUAL_CONFIG["title"] = "Manual Run"
MANUAL_CONFIG["save_path_folder"] = os.path.join(os.getcwd(), "manual_logs")
MANUAL_CONFIG["interval_datastorage"] = 1
MANUAL_CONFIG["interval_episode_log"] = 1
