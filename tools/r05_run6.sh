T="tests/test_hip_car_parity.py::test_set_state_that_rewinds_the_episode_does_not_reuse_an_overwritten_walk"
timeout 300 python -m pytest "$T" -x -q -m gpu 2>&1 | tail -3
CRL_LIB_VARIANT=abl CRL_CAR_ABL_KEEP_TAG=1 timeout 300 python -m pytest "$T" -x -q -m gpu 2>&1 | tail -5
