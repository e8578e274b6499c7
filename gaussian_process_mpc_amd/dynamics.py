class Dynamics:  # placeholder
    pass
