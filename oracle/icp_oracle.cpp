// placeholder replaced below
