Simulation=None
