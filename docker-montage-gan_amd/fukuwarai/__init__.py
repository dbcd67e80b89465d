"""Layer position estimator (spatial transformer) of the global generator (reference fukuwarai/)."""
