def mean_prop_torch(*a, **k): raise NotImplementedError
def variance_prop_torch(*a, **k): raise NotImplementedError
def covariance_prop_torch(*a, **k): raise NotImplementedError
