"""Fixture G14: what each variant does to a constructed environment before reset() (data shared by
tests/golden/make_golden.py, which applies it to the reference, and by the tests, which apply it to the
oracle and to the drop-in)."""

G14_VARIANTS = {
    "no_microclimate": dict(call="set_use_microclimate", args=(False,)),
    "slow_time": dict(attrs=dict(dt=0.5, agent_gamma=0.1)),
    "other_physics": dict(attrs=dict(q2=3.5273e9 / 4.0, temp_optimal=290.0, gamma=0.3, g=0.004, albedo_bare=0.45)),
    "wide_albedo": dict(attrs=dict(albedo_light=0.9, albedo_dark=0.1, dt=2.0)),
}
