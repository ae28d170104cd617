"""Persistence container (row f3), with the CPU oracle as the scheme: behaviours of the
reference's tests/test_containers.py that do not depend on xarray."""
import numpy as np
import pytest

from oracle import numpy_path as ora
from triflow_amd import Model, Simulation, retrieve_container


def make_sim(**kw):
    m = Model("k * dxxT", "T", "k", compiler=ora.numpy_compiler)
    x = np.linspace(0, 10, 50, endpoint=False)
    fields = m.fields_template(x=x, T=np.cos(x * 2 * np.pi / 10))
    return Simulation(m, fields, dict(periodic=True, k=1), dt=.5, tmax=2., scheme=ora.Theta,
                      time_stepping=False, **kw)


def test_in_memory_all_and_last():
    sim = make_sim()
    c = sim.attach_container()
    sim.run(progress=False)
    assert c.data["T"].shape == (5, 50)            # initial state + 4 steps
    assert np.allclose(c.data["t"], [0, .5, 1., 1.5, 2.])
    assert np.array_equal(c.data["T"][-1], np.asarray(sim.fields["T"]))
    sim = make_sim()
    c = sim.attach_container(save="last")
    sim.run(progress=False)
    assert c.data["T"].shape == (1, 50) and c.data["t"][0] == 2.


def test_on_disk_chunks_merge_and_retrieve(tmp_path):
    sim = make_sim(id="run")
    c = sim.attach_container(str(tmp_path), nbuffer=2)
    sim.run(progress=False)
    back = retrieve_container(str(tmp_path / "run"))
    assert np.array_equal(back.data["T"], c.data["T"])
    assert back.metadata["k"] == 1 and back.metadata["periodic"] is True
    last = retrieve_container(str(tmp_path / "run"), isel="last")
    assert last.data["T"].shape == (50,)
    with pytest.raises(FileExistsError):
        make_sim(id="run").attach_container(str(tmp_path))
    make_sim(id="run").attach_container(str(tmp_path), force=True)
    with pytest.raises(FileNotFoundError):
        retrieve_container(str(tmp_path / "nothing"))
    with pytest.raises(ValueError):
        make_sim().attach_container(save="some")
