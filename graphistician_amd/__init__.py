"""graphistician_amd — MI355X-native Gibbs engine behind graphistician's
networks.py surface (resample / log_likelihood).  See DESIGN.md."""
