"""`nvit.kohonen` of the reference (nvit/kohonen.py), served by the MI355X implementation.  See nvit_amd/kohonen.py."""
from nvit_amd.kohonen import KohonenMap  # noqa: F401
