"""Price tables for the Ticker env (reference: fed_gym/envs/data/).  No market data ships with this package: put
`<ticker>.csv` files (Yahoo-style columns Open, Close, Volume) here or point GOLDSRL_DATA_DIR at a directory."""
