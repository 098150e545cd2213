"""Host logic: the reference's task YAML (Hydra/OmegaConf interpolations, train.py:58-63) resolved without Hydra."""
import os
import pytest

from isaacgymdyros_amd.config import default_cfg, load_task_yaml

SNIPPET = '''
name: DyrosDynamicWalk
physics_engine: ${..physics_engine}
env:
  numEnvs: ${resolve_default:4096,${...num_envs}}
  controlFrequencyInv: 2
sim:
  dt: 0.002
  use_gpu_pipeline: ${eq:${...pipeline},"gpu"}
  physx:
    num_threads: ${....num_threads}
    use_gpu: ${contains:"cuda",${....sim_device}}
    num_subscenes: ${....num_subscenes}
'''


def test_resolvers_follow_the_reference_semantics():
    c = load_task_yaml(SNIPPET)
    assert c["physics_engine"] == "physx" and c["env"]["numEnvs"] == 4096
    assert c["sim"]["use_gpu_pipeline"] is True and c["sim"]["physx"]["use_gpu"] is True
    assert c["sim"]["physx"]["num_threads"] == 4 and c["sim"]["physx"]["num_subscenes"] == 4
    c = load_task_yaml(SNIPPET, num_envs=16384, pipeline="cpu", sim_device="cpu")
    assert c["env"]["numEnvs"] == 16384 and c["sim"]["use_gpu_pipeline"] is False and c["sim"]["physx"]["use_gpu"] is False
    with pytest.raises(KeyError):
        load_task_yaml("a: ${..no_such_key}")


def test_reference_yaml_matches_the_built_in_defaults():
    path = "/root/reference/python/IsaacGymEnvs/isaacgymenvs/cfg/task/DyrosDynamicWalk.yaml"
    if not os.path.exists(path):
        pytest.skip("reference checkout not present (GPU box)")
    ref, mine = load_task_yaml(path), default_cfg(4096)
    for sec in ("env", "sim"):
        for k, v in ref[sec].items():
            if k in mine[sec] and not isinstance(v, dict):
                assert mine[sec][k] == v, (sec, k, mine[sec][k], v)
    assert ref["sim"]["physx"]["num_position_iterations"] == mine["sim"]["physx"]["num_position_iterations"]
    assert ref["task"]["randomize"] == mine["task"]["randomize"]
    ra = ref["task"]["randomization_params"]["actor_params"]["humanoid"]
    ma = mine["task"]["randomization_params"]["actor_params"]["humanoid"]
    assert ra["dof_properties"]["damping"]["range"] == ma["dof_properties"]["damping"]["range"]
    assert ra["rigid_body_properties"]["mass"]["range"] == ma["rigid_body_properties"]["mass"]["range"]
