class RiskSensitiveMPC:  # placeholder
    pass
